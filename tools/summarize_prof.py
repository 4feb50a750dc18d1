#!/usr/bin/env python
"""Aggregate rocprofv3 CSV output (kernel stats + PMC passes) per kernel name -> JSON summary for profiles/."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"conv_mfma_kernel<([^>]*)>", name)
    if m:
        return "conv_mfma_kernel<%s>" % m.group(1).replace("(anonymous namespace)::", "").replace(" ", "")
    m = re.search(r"(conv_f16_kernel)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    m = re.search(r"(conv_wino\w*_kernel)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*", "", name)[:80]


def find(d, pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return fs[0] if fs else None


def kernel_stats(d):
    f = find(d, "*kernel_stats.csv")
    out = {}
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        out[short(r["Name"])] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                                 "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
    return out


def pmc(d):
    f = find(d, "*counter_collection.csv")
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return {k: dict(v, dispatches=len(cnt[k])) for k, v in agg.items()}


def main():
    root = sys.argv[1]
    res = {"kernel_stats": kernel_stats(os.path.join(root, "trace"))}
    for name in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
        res[name] = pmc(os.path.join(root, name))
    # derived, for the dominant kernel
    for k, v in res["pmc_sq"].items():
        if v.get("SQ_BUSY_CYCLES"):
            v["mfma_busy_frac_of_gui_active"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, v.get("GRBM_GUI_ACTIVE", 0)) / 1024.0
    hbm = {}
    for k in res["pmc_fetch"]:
        fe = res["pmc_fetch"][k]
        wr = res["pmc_write"].get(k, {})
        n = max(1, fe.get("dispatches", 1))
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md HBM section)
        hbm[k] = {"fetch_bytes_per_launch_corrected": fe.get("FETCH_SIZE", 0) * 1024 * 2 / n,
                  "write_bytes_per_launch": wr.get("WRITE_SIZE", 0) * 1024 / max(1, wr.get("dispatches", 1))}
    res["hbm"] = hbm
    # drop torch's own tiny kernels from the per-kernel tables (keep ours)
    for sec in ("kernel_stats", "pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "hbm"):
        res[sec] = {k: v for k, v in res[sec].items() if not k.startswith(("void at::", "__amd_rocclr"))}
    json.dump(res, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
