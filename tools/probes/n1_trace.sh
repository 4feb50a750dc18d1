cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_n1; mkdir -p $R/gpurun_out/prof_n1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_n1 -o t --output-format csv -- python $R/tools/probes/n1_trace.py > $R/gpurun_out/prof_n1/log.txt 2>&1
f=$(find $R/gpurun_out/prof_n1 -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward: take the last 60 launches, print name / grid / duration / gap to the previous end
tail = rows[-60:]
prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev is None else s - prev
    prev = e
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    print(f"{name:50s} grid {int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])):6d} wg {int(r['Workgroup_Size_X']):4d}  {(e-s)/1000:7.1f} us  gap {gap/1000:6.1f}")
PY
