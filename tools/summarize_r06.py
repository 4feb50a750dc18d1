#!/usr/bin/env python
"""rocprofv3 passes of bench.py (tools/rocprof_passes.sh) -> per launch GROUP summary: the conv kernel names do not carry the channel
count, so dispatches are split by (kernel, workgroups).  Writes <out>/rocprof_summary.json and <out>/pmc_latest.json (the dominant group:
conv_wx4_kernel<3,*> at the 96-channel grid of the 32x256x256 workload: 4096 workgroups), which bench.py reads for roofline.traffic.

usage: tools/summarize_r06.py gpurun_out/prof_r06 profiles r06"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def short(n):
    m = re.search(r"(conv_wx4h_kernel|conv_wx4_kernel|conv_exit_kernel|conv_entry_kernel|knet_body_kernel|conv_f16\w*_kernel|conv_mfma_kernel|conv_wino_row_kernel|conv_wgrad_kernel|pack_\w+_kernel|conv3x3_thin_kernel)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")).replace(" ", "") if m else None


def find(d, pat):
    fs = glob.glob(os.path.join(src, d, "**", pat), recursive=True)
    return fs[0] if fs else None


trace = collections.defaultdict(list)
for r in csv.DictReader(open(find("trace", "*kernel_trace.csv"))):
    k = short(r["Kernel_Name"])
    if k:
        trace[(k, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
groups = {}
for (k, wgs), v in trace.items():
    groups[f"{k} @ {wgs} workgroups"] = {"launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "total_ms": sum(v) / 1e6}
pmc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    f, t = find(d, "*counter_collection.csv"), find(d, "*kernel_trace.csv")
    if not f or not t:
        continue
    grid = {r["Dispatch_Id"]: int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) for r in csv.DictReader(open(t))}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        key = f"{k} @ {grid.get(r['Dispatch_Id'], 0)} workgroups"
        pmc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]].add(r["Dispatch_Id"])
for key, c in pmc.items():
    g = groups.setdefault(key, {})
    per = {name: val / max(1, len(cnt[key][name])) for name, val in c.items()}
    g["pmc_per_launch"] = per
    if per.get("GRBM_GUI_ACTIVE"):
        g["mfma_busy_frac_of_simd_cycles"] = per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (per["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "FETCH_SIZE" in per or "WRITE_SIZE" in per:
        # KiB counters; gfx950 FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated
        g["hbm_fetch_bytes_corrected"] = per.get("FETCH_SIZE", 0) * 1024 * 2
        g["hbm_write_bytes"] = per.get("WRITE_SIZE", 0) * 1024
json.dump(dict(sorted(groups.items(), key=lambda kv: -kv[1].get("total_ms", 0))), open(os.path.join(out, f"{tag}_rocprof_summary.json"), "w"), indent=1)
# dominant group: the 96-channel conv_wx4 launches (4096 workgroups at 32x256x256), conv1-type (EPI 0) and conv2-type (EPI 1, residual)
dom = {k: v for k, v in groups.items() if k.startswith("conv_wx4_kernel<3,") and k.endswith("@ 4096 workgroups") and v.get("launches")}
if dom:
    n = sum(v["launches"] for v in dom.values())
    fetch = sum(v.get("hbm_fetch_bytes_corrected", 0) * v["launches"] for v in dom.values()) / n
    write = sum(v.get("hbm_write_bytes", 0) * v["launches"] for v in dom.values()) / n
    busy = sum(v.get("mfma_busy_frac_of_simd_cycles", 0) * v["launches"] for v in dom.values()) / n
    avg = sum(v["avg_us"] * v["launches"] for v in dom.values()) / n
    mops = sum(v.get("pmc_per_launch", {}).get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0) * v["launches"] for v in dom.values()) / n
    json.dump({"kernel": "conv_wx4_kernel<3,EPI,PRE> (96-channel launch group: 4096 workgroups at 32x256x256)", "form": "wx4",
               "mfma_mops_f16_per_launch": mops, "mfma_flop_executed_per_launch": mops * 512,
               "source": f"profiles/{tag}_rocprof_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_*, separate passes, bench.py 256x256 x32)",
               "members": sorted(dom), "fetch_bytes_per_launch": fetch,
               "fetch_correction": "FETCH_SIZE KiB x1024 x2 (gfx950 counts 128-B requests as 64 B; MI355X_MICROARCH.md HBM section)",
               "write_bytes_per_launch": write, "write_correction": "WRITE_SIZE KiB x1024 (uncalibrated)",
               "hbm_bytes_per_launch": fetch + write,
               "algorithmic_bytes_per_launch": "805 MB input + 805 MB output (+805 MB residual on the conv2-type launches, half of them)",
               "mfma_busy_frac_of_simd_cycles": busy, "avg_launch_us_kernel_trace": avg},
              open(os.path.join(out, "pmc_latest.json"), "w"), indent=1)
    print("dominant group:", n, "launches, avg", round(avg, 1), "us, hbm", round((fetch + write) / 1e9, 3), "GB, mfma busy", round(busy, 3))
for k, v in list(sorted(groups.items(), key=lambda kv: -kv[1].get("total_ms", 0)))[:12]:
    print("%-58s %4d x %8.1f us  busy %s" % (k, v.get("launches", 0), v.get("avg_us", 0), round(v.get("mfma_busy_frac_of_simd_cycles", 0), 3)))
