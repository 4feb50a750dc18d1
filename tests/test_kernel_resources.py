"""Register / scratch gate of the hot translation units (VERDICT r04 next #5a): no kernel of the convolution path may touch scratch memory,
with the exceptions listed (and bounded) below.  The numbers are the compiler's own (`-Rpass-analysis=kernel-resource-usage`), written
beside every object by virnet_amd/csrc/Makefile (build/csrc/<unit>.kres); a unit whose remarks are missing or older than its source is
compiled here for the remarks alone (device pass only, ~1 min for the largest unit) -- `tools/kres.py <source>` prints the same table."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "virnet_amd", "csrc")
HOT = ["conv_f16_wx4", "conv_f16_wx4p", "conv_f16_wx4h", "conv_f16", "conv_f16_s2", "conv_f16_pw", "conv_exit", "conv_entry", "wgrad_f16", "knet_body"]

# (unit, kernel regex) -> scratch bytes per lane tolerated, with the reason.  Everything else: zero.
ALLOWED = [
    # the 8-row Winograd form with THREE slabs and SFT staging: picked only for launches of at most one workgroup per CU (single-image
    # SISR) or through VIRNET_WX4_ROWS=8 (its 80 KB of LDS + the SFT table: one workgroup per CU); one register is parked once per tile
    # OUTSIDE the K loop
    ("conv_f16_wx4h", r"conv_wx4h_kernel<3, \d, 2, 0>", 8),
    # the direct kernel's 2 x 3-block tile sits at the 256-register budget: two fragment offsets (round 4: 6-12 values)
    ("conv_f16", r"conv_f16_kernel<2, 3, \d, 0, [01], 0>", 16),
    # transposed conv, two-chunk stages at 128 registers: two register pairs parked in the EPILOGUE (stored and re-loaded once per tile)
    ("conv_f16_pw", r"conv_f16_pw_kernel<2, 3, 2>", 20),
    # seven-slab stride-2 form (SISR 224 channels): SGPR spills only (no scratch instruction in the kernel; the bytes are the lanes' save area)
    ("conv_f16_s2", r"conv_f16_s2_kernel<1, 7>", 20),
]


def _remarks(unit):
    src = os.path.join(CSRC, unit + ".hip")
    kres = os.path.join(ROOT, "build", "csrc", unit + ".kres")
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    if os.path.exists(kres) and os.path.getmtime(kres) >= max(os.path.getmtime(d) for d in deps):
        with open(kres) as f:
            return f.read()
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc and no build/csrc/%s.kres from the build" % unit)
    return subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-c", src, "-o", "/dev/null", "--cuda-device-only",
                           "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=1200).stderr


def _table(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|ScratchSize \[bytes/lane\]|VGPRs Spill|VGPRs|AGPRs):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = int(v)
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n") if rows else []
    for r, n in zip(rows, names):
        r["pretty"] = n.replace("(anonymous namespace)::", "").split("(")[0]
    return rows


@pytest.mark.parametrize("unit", HOT)
def test_hot_kernels_do_not_use_scratch(unit):
    rows = _table(_remarks(unit))
    assert rows, f"no kernel-resource remarks for {unit}"
    bad = []
    for r in rows:
        limit = 0
        for u, pat, lim in ALLOWED:
            if u == unit and re.search(pat, r["pretty"]):
                limit = lim
        if r.get("ScratchSize [bytes/lane]", 0) > limit:
            bad.append((r["pretty"], r.get("ScratchSize [bytes/lane]"), r.get("VGPRs Spill"), limit))
    assert not bad, f"{unit}: kernels over their scratch budget (name, bytes/lane, spilled VGPRs, allowed): {bad}"


def test_winograd_kernels_of_the_metric_have_no_spilled_vgpr():
    """every instantiation of the 16-row Winograd kernel (all PRE / EPI forms, SFT staging included) and every 8-row instantiation but the
    three-slab SFT form (ALLOWED above: one register parked once per tile outside the K loop; the launcher picks it for launches of at most
    one workgroup per CU only): zero spilled vector registers"""
    for unit, skip in (("conv_f16_wx4", None), ("conv_f16_wx4h", r"conv_wx4h_kernel<3, \d, 2, 0>")):
        for r in _table(_remarks(unit)):
            if "conv_wx4" not in r["pretty"] or (skip and re.search(skip, r["pretty"])):
                continue
            assert r.get("VGPRs Spill", 0) == 0 and r.get("ScratchSize [bytes/lane]", 0) == 0, r
