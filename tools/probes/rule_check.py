#!/usr/bin/env python
"""Does the launch-shape rule pick the fastest kernel form?  

For the single-image call pattern of the reference's scripts (scripts/denoising_virnet_syn.py:133-134, scripts/testing_demo.py:87-93) at the
sizes they meet -- 481 x 321 (CBSD68), 500 x 500 (McMaster), 256 x 256, 128 x 128 -- every res-block convolution of the three levels is
timed in each form the library could take (Winograd 8-row tiles, Winograd 16-row tiles, the direct split-fp16 kernel; interleaved rounds,
K launches per sample, median over the rounds) and beside them what the default rule takes (ops.wx4_shape_ok + virnet_conv_wx4's tile-form
rule + the slab groupings).  Prints one row per (size, level, launch type) and exits 1 if the rule's pick is more than --tol (3 %) + 0.5 us
slower than the best candidate.

    python tools/probes/rule_check.py [--sizes 256x256,128x128,481x321,500x500] [--json out.json]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from virnet_amd import ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402

KNOBS = ("VIRNET_CONV_FORM", "VIRNET_WX4_ROWS", "VIRNET_WX4_MIN_WGS", "VIRNET_WX4_MIN_SLAB_WGS", "VIRNET_WX4_MIN_TILES", "VIRNET_WX4_MIN_FILL", "VIRNET_WX4_NREP")
FORCE = {"VIRNET_WX4_MIN_WGS": "1", "VIRNET_WX4_MIN_SLAB_WGS": "0", "VIRNET_WX4_MIN_TILES": "0", "VIRNET_WX4_MIN_FILL": "0"}
CANDS = {
    "rule": {},
    "wx4 8-row": dict(FORCE, VIRNET_CONV_FORM="wx4", VIRNET_WX4_ROWS="8"),
    "wx4 16-row": dict(FORCE, VIRNET_CONV_FORM="wx4", VIRNET_WX4_ROWS="16"),
    "direct f16x3": dict(VIRNET_CONV_FORM="f16x3"),
}


def set_env(d):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="256x256,128x128,481x321,500x500")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--tol", type=float, default=0.03)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    torch.manual_seed(0)
    rows, bad = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for size in a.sizes.split(","):
        H, W = (int(v) for v in size.split("x"))
        Hp, Wp = (H + 3) // 4 * 4, (W + 3) // 4 * 4                     # AttResUNet pads to multiples of 2^(depth-1) (util_net.py:20-25)
        for lvl, c in enumerate((96, 192, 288)):
            h, w = Hp >> lvl, Wp >> lvl
            cp = ConvParam(c, c, 3).cuda()
            x = torch.rand(1, h, w, c, device="cuda") - 0.5
            res = torch.rand(1, h, w, c, device="cuda") - 0.5
            for mode, kw in (("conv1", dict(in_slope=0.2, want_raw=False, want_act=True)), ("conv2", dict(res=res, want_raw=True))):
                # every candidate as a captured graph of K launches: the kernels are 15-90 us, the host needs ~10 us per eager launch --
                # replayed, the interval between the two events is the kernels' own time
                graphs, taken = {}, {}
                for name, env in CANDS.items():
                    set_env(env)
                    pw = cp.packed()
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            ops.conv_mfma(x, pw, **kw)
                    torch.cuda.current_stream().wait_stream(side)
                    if name == "rule":
                        taken = ops.wx4_last_plan() if ops.wx4_shape_ok(1, h, w, c) else {"rows": 0}
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for _ in range(a.k):
                            ops.conv_mfma(x, pw, **kw)
                    graphs[name] = g
                samples = {name: [] for name in CANDS}
                for _ in range(a.rounds):
                    for name in CANDS:
                        e0.record()
                        graphs[name].replay()
                        e1.record()
                        e1.synchronize()
                        samples[name].append(e0.elapsed_time(e1) / a.k * 1e3)
                med = {n: sorted(v)[len(v) // 2] for n, v in samples.items()}
                best = min((v, n) for n, v in med.items() if n != "rule")
                ok = med["rule"] <= best[0] * (1 + a.tol) + 0.5
                what = "direct" if not taken.get("rows") else f"wx4 {taken['rows']}-row x{taken['slabs']} slab(s), {taken['launches']} launch(es)"
                rows.append({"size": size, "level": lvl, "channels": c, "hw": [h, w], "launch": mode, "rule_takes": what, "us": {n: round(v, 2) for n, v in med.items()},
                             "best": best[1], "ok": ok})
                print(f"{size:8s} l{lvl} {c:3d}ch {h:3d}x{w:<3d} {mode}: rule {med['rule']:6.1f} us ({what}) | " +
                      " | ".join(f"{n} {med[n]:6.1f}" for n in CANDS if n != "rule") + f" | best {best[1]}" + ("" if ok else "   <-- RULE MISSES"), flush=True)
                if not ok:
                    bad.append(rows[-1])
    set_env({})
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)
    print(f"{len(rows)} launch shapes, rule within {a.tol * 100:.0f} % + 0.5 us of the best form on {len(rows) - len(bad)}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
