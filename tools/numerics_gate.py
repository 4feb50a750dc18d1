"""Numerics gate for the algorithmic levers on the conv path, run on the CPU before any kernel is written.

The whole denoise-syn network (scripts/denoising_virnet_syn.py:62-71 configuration, synthetic weights) is evaluated with every
stride-1 3x3 convolution of >= 32 input channels replaced by an EMULATION of a candidate arithmetic and compared against an fp64
run of the same network.  The emulations are exact models of what the matrix pipe would compute up to summation order:

  f32        plain fp32 (what v_mfma_f32_32x32x2_f32 computes; the direct kernel)
  wino2      Winograd F(2x2,3x3), fp32 transforms and products (the shipped kernel form)
  wino4      Winograd F(4x4,3x3), fp32 (36/144 multiplies; Lavin & Gray matrices)
  wino2_f16x3 / wino4_f16x3   the same with the position products on the f16 pipe (split operands, three products)
  bf16x3     operands split x = hi + lo in bf16; products hi*hi + hi*lo + lo*hi accumulated in fp32
  bf16x6     three-way bf16 split, the six products of order <= 2^-16
  f16x3      operands split in fp16 (weights pre-scaled by a power of two per layer); hi*hi + hi*lo + lo*hi in fp32
  f16x4      the same plus lo*lo
  f16x3_ftz  f16x3 with fp16 subnormals flushed to zero (what a pipe without fp16 denormal support would do)
  wx4        the shipped kernel (conv_f16_wx4.hip): F(4,3) along x, direct along y, split-fp16 position products
  wx6        F(6,3) along x, direct along y (8 positions per 6 outputs: 1.33 executed FLOP per algorithmic one), split-fp16 products
  w42        the MIXED 2-D form VERDICT r05 next #6 asks to be priced: F(4,3) along x and F(2,3) along y, split-fp16 position products
             (6 x 4 = 24 positions per 4 x 2 outputs: 1.0 executed FLOP per algorithmic FLOP with three products, against wx4's 1.5)

A product of two 11-bit (fp16) or 8-bit (bf16) significands is exact in fp32, so conv2d in fp32 over the split operands reproduces
the MFMA result except for the order of the fp32 additions.

Usage: python tools/numerics_gate.py [--size 96] [--batch 1] [--modes f32,wino2,...]   (writes a markdown table to stdout)
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import cpu_ref  # noqa: E402  (tools/ is test infrastructure, like tests/)
from virnet_amd.utils.synth import synth_images, synth_state_dict  # noqa: E402

_REAL_CONV = F.conv2d


def split_bf16(x, parts):
    out, r = [], x
    for _ in range(parts):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def split_f16(x, parts, ftz=False):
    out, r = [], x
    for _ in range(parts):
        p = r.to(torch.float16).to(torch.float32)
        if ftz:
            p = torch.where(p.abs() < 6.103515625e-05, torch.zeros_like(p), p)
        out.append(p)
        r = r - p
    return out


def pow2_scale(w, target=16384.0):
    m = float(w.abs().max())
    return 2.0 ** math.floor(math.log2(target / m)) if m > 0 else 1.0


def wino_mats(m):
    if m == 1:      # "F(1,3)": the direct form written as a transform (identity in, three taps summed out)
        eye = torch.eye(3, dtype=torch.float64)
        return eye, eye, torch.ones(1, 3, dtype=torch.float64)
    if m == 6:      # F(6,3), points 0, +-1, +-2, +-1/2, inf (Lavin & Gray): 8 positions per 6 outputs
        BT = torch.tensor([[1, 0, -21 / 4, 0, 21 / 4, 0, -1, 0], [0, 1, 1, -17 / 4, -17 / 4, 1, 1, 0], [0, -1, 1, 17 / 4, -17 / 4, -1, 1, 0],
                           [0, 1 / 2, 1 / 4, -5 / 2, -5 / 4, 2, 1, 0], [0, -1 / 2, 1 / 4, 5 / 2, -5 / 4, -2, 1, 0],
                           [0, 2, 4, -5 / 2, -5, 1 / 2, 1, 0], [0, -2, 4, 5 / 2, -5, -1 / 2, 1, 0], [0, -1, 0, 21 / 4, 0, -21 / 4, 0, 1]], dtype=torch.float64)
        G = torch.tensor([[1, 0, 0], [-2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45], [1 / 90, -1 / 45, 2 / 45],
                          [32 / 45, 16 / 45, 8 / 45], [32 / 45, -16 / 45, 8 / 45], [0, 0, 1]], dtype=torch.float64)
        AT = torch.tensor([[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 1 / 2, -1 / 2, 0], [0, 1, 1, 4, 4, 1 / 4, 1 / 4, 0],
                           [0, 1, -1, 8, -8, 1 / 8, -1 / 8, 0], [0, 1, 1, 16, 16, 1 / 16, 1 / 16, 0], [0, 1, -1, 32, -32, 1 / 32, -1 / 32, 1]], dtype=torch.float64)
        return BT, G, AT
    if m == 2:
        BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
        G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
        AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    else:
        BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                           [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
        G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                          [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
        AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
    return BT, G, AT


def conv_wino(x, w, m, split=False):
    """F(m x m, 3x3) with every step in fp32 (weights transformed in fp64 then rounded once, as the packing kernel does).
    ``split``: the position products on the f16 pipe with split operands (U pre-scaled by a power of two), as conv_f16 does for the
    direct form -- the candidate that executes 3*16/36 FLOPs per algorithmic FLOP."""
    BT, G, AT = wino_mats(m)
    a = m + 2
    n, c, h, wd = x.shape
    co = w.shape[0]
    hp, wp = -(-h // m) * m, -(-wd // m) * m
    xp = F.pad(x, (1, 1 + wp - wd, 1, 1 + hp - h))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                       # [n, c, th, tw, a, a]
    BTf, ATf = BT.float(), AT.float()
    V = torch.einsum("ij,nctujk,lk->nctuil", BTf, tiles, BTf)         # fp32 transform
    U = torch.einsum("ij,ocjk,lk->ocil", G, w.double(), G).float()    # [co, c, a, a]
    if split:
        sc = pow2_scale(U)
        (uh, ul), (vh, vl) = split_f16(U * sc, 2), split_f16(V, 2)
        M = (torch.einsum("ocil,nctuil->notuil", ul, vh) + torch.einsum("ocil,nctuil->notuil", uh, vl)
             + torch.einsum("ocil,nctuil->notuil", uh, vh)) * (1.0 / sc)
    else:
        M = torch.einsum("ocil,nctuil->notuil", U, V)                 # fp32 products / accumulation
    Y = torch.einsum("ij,notujk,lk->notuil", ATf, M, ATf)             # [n, co, th, tw, m, m]
    th, tw = Y.shape[2], Y.shape[3]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, co, th * m, tw * m)
    return y[..., :h, :wd]


def conv_wino_xy(x, w, my, mx, split=True):
    """F(my,3) along y x F(mx,3) along x (m = 1: direct along that axis), fp32 input / output transforms, weights transformed in fp64 and
    rounded once, position products with split-fp16 operands (three products, fp32 accumulation) -- conv_wino generalised per axis."""
    BTy, Gy, ATy = wino_mats(my)
    BTx, Gx, ATx = wino_mats(mx)
    ay, ax = my + 2, mx + 2
    n, c, h, wd = x.shape
    co = w.shape[0]
    hp, wp = -(-h // my) * my, -(-wd // mx) * mx
    xp = F.pad(x, (1, 1 + wp - wd, 1, 1 + hp - h))
    tiles = xp.unfold(2, ay, my).unfold(3, ax, mx)                   # [n, c, th, tw, ay, ax]
    V = torch.einsum("ij,nctujk,lk->nctuil", BTy.float(), tiles, BTx.float())
    U = torch.einsum("ij,ocjk,lk->ocil", Gy, w.double(), Gx).float()
    if split:
        sc = pow2_scale(U)
        (uh, ul), (vh, vl) = split_f16(U * sc, 2), split_f16(V, 2)
        M = (torch.einsum("ocil,nctuil->notuil", ul, vh) + torch.einsum("ocil,nctuil->notuil", uh, vl)
             + torch.einsum("ocil,nctuil->notuil", uh, vh)) * (1.0 / sc)
    else:
        M = torch.einsum("ocil,nctuil->notuil", U, V)
    Y = torch.einsum("ij,notujk,lk->notuil", ATy.float(), M, ATx.float())
    th, tw = Y.shape[2], Y.shape[3]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, co, th * my, tw * mx)
    return y[..., :h, :wd]


def make_conv(mode):
    def conv(x, w, b=None, stride=1, padding=0, *a, **k):
        elig = (stride == 1 and w.shape[-1] == 3 and w.shape[1] >= 32 and w.shape[0] % 32 == 0 and x.dtype == torch.float32)
        if not elig or mode == "f32":
            return _REAL_CONV(x, w, b, stride, padding, *a, **k)
        if mode in ("wino2", "wino4", "wino2_f16x3", "wino4_f16x3"):
            y = conv_wino(x, w, 2 if mode.startswith("wino2") else 4, split=mode.endswith("f16x3"))
        elif mode in ("wx4", "w42", "wx4_f32", "w42_f32"):
            y = conv_wino_xy(x, w, 1 if mode.startswith("wx4") else 2, 4, split=not mode.endswith("f32"))
        elif mode in ("wx6", "wx6_f32"):      # F(6,3) along x, direct along y: 24 instead of 27 k-steps per 6 output pixels (1.33 executed per algorithmic FLOP)
            y = conv_wino_xy(x, w, 1, 6, split=not mode.endswith("f32"))
        elif mode in ("bf16x3", "bf16x6"):
            parts = 2 if mode == "bf16x3" else 3
            xs, ws = split_bf16(x, parts), split_bf16(w, parts)
            y = 0
            order = sorted(((i, j) for i in range(parts) for j in range(parts) if i + j < parts), key=lambda t: -(t[0] + t[1]))
            for i, j in order:                                           # small terms first
                y = y + _REAL_CONV(xs[i], ws[j], None, 1, padding)
        elif mode in ("f16x3", "f16x4", "f16x3_ftz", "f16x3_noscale"):
            ftz = mode.endswith("ftz")
            s = 1.0 if mode.endswith("noscale") else pow2_scale(w)
            xs, ws = split_f16(x, 2, ftz), split_f16(w * s, 2, ftz)
            y = _REAL_CONV(xs[1], ws[0], None, 1, padding) + _REAL_CONV(xs[0], ws[1], None, 1, padding)
            if mode == "f16x4":
                y = y + _REAL_CONV(xs[1], ws[1], None, 1, padding)
            y = (y + _REAL_CONV(xs[0], ws[0], None, 1, padding)) * (1.0 / s)
        else:
            raise ValueError(mode)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return conv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--modes", default="f32,wino2,wino4,bf16x3,bf16x6,f16x3,f16x4,f16x3_ftz,f16x3_noscale")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    with open(os.path.join(REPO, "tests", "golden", "manifest.json")) as f:
        man = json.load(f)
    cfg = dict(man["configs"]["syn"])
    cfg.pop("kind")
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn")}
    sd = synth_state_dict({k: tuple(s) for k, s in man["shapes"]["syn"].items()})
    x = synth_images(args.batch, 3, args.size, args.size)
    t0 = time.time()
    sd64 = {k: v.double() for k, v in sd.items()}
    mu64, sg64 = cpu_ref.virnet_denoise(sd64, x.double(), **kw)
    print(f"fp64 reference: {time.time() - t0:.1f} s, |mu| max {float(mu64.abs().max()):.2f}, sigma max {float(sg64.max()):.3g}",
          file=sys.stderr)
    rows = []
    for mode in args.modes.split(","):
        t0 = time.time()
        cpu_ref.F.conv2d = make_conv(mode)
        try:
            mu, sg = cpu_ref.virnet_denoise(sd, x, **kw)
        finally:
            cpu_ref.F.conv2d = _REAL_CONV
        e_mu = float((mu.double() - mu64).abs().max())
        r_mu = float((mu.double() - mu64).pow(2).mean().sqrt())
        e_sg = float(((sg.double() - sg64).abs() / sg64.abs().clamp_min(1e-30)).max())
        rows.append({"mode": mode, "mu_max_abs": e_mu, "mu_rms": r_mu, "sigma_max_rel": e_sg, "seconds": time.time() - t0})
        print(f"{mode:14s} mu max-abs {e_mu:.3e}  rms {r_mu:.3e}  sigma max-rel {e_sg:.3e}  ({time.time() - t0:.1f} s)", file=sys.stderr)
    print(f"| arithmetic | max-abs error on mu vs fp64 | rms | max-rel error on sigma |")
    print("|---|---|---|---|")
    for r in rows:
        print(f"| {r['mode']} | {r['mu_max_abs']:.2e} | {r['mu_rms']:.2e} | {r['sigma_max_rel']:.2e} |")
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"size": args.size, "batch": args.batch, "mu_absmax": float(mu64.abs().max()), "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
