"""What a TWO-term weight gradient (dY rounded to fp16, no dY_lo x A_hi product) would cost in accuracy -- VERDICT r04 next #6's hint
("run the lo x lo-free two-term form only where test_wgrad_f16_gpu.py's 1e-4 bar still holds").  CPU / numpy, no GPU needed: the
contraction dW[co][ci] = sum_p dY[p][co] * A[p][ci] of csrc/wgrad_f16.hip with dY's lo plane dropped, against fp64.
Prints max|err| / max|dW| for (a) noise-like gradients (dY independent of A: the case the parity tests draw) and (b) a gradient with a
coherent component.  python tools/probes/two_term_wgrad_error.py"""
import numpy as np

rng = np.random.default_rng(0)
for n in (6720, 131072, 524288):                    # pixels contracted: the parity test's 4 x 24 x 70, one level-1 / level-0 training launch
    a = rng.standard_normal((n, 96)).astype(np.float32)
    a = np.where(a > 0, a, 0.2 * a)
    dy = rng.standard_normal((n, 96)).astype(np.float32)
    for name, g in (("noise-like", dy), ("coherent (dY + 0.3 A)", (dy + 0.3 * a).astype(np.float32))):
        ref = g.astype(np.float64).T @ a.astype(np.float64)
        two = g.astype(np.float16).astype(np.float64).T @ a.astype(np.float64)
        print(f"pixels {n:7d}  {name:24s} max|err| / max|dW| = {np.abs(two - ref).max() / np.abs(ref).max():.2e}")
