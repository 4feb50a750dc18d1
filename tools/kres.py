#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy summary of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), gfx950.

usage: tools/kres.py virnet_amd/csrc/conv_f16.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-c", src, "-o", "/dev/null",
                      "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|"
                  r"VGPRs Spill|LDS Size \[bytes/block\]):\s*(\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    print("%-56s vgpr %4s agpr %3s sgpr %4s scratch %4s occ %s spill v%s s%s" % (
        name, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
        r.get("VGPRs Spill"), r.get("SGPRs Spill")))
