#!/bin/bash
# round 6: effective shader clock of the per-item and the persistent form of the 96-channel launch under sustained load
# (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration over the last 200 of 500 back-to-back launches; PMC pass of its own, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for P in 0 1; do
  rm -rf $R/gpurun_out/clk_p$P
  VIRNET_WX4_PERSIST=$P rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/clk_p$P -o p --output-format csv -- \
    python $R/tools/bench_conv.py --shapes l0 --mode pre --iters 500 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
d="$R/gpurun_out/clk_p$P"
f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)[0]
t=glob.glob(d+"/**/*kernel_trace.csv",recursive=True)[0]
dur={}
for r in csv.DictReader(open(t)):
    if "conv_wx4" in r["Kernel_Name"]: dur[int(r["Dispatch_Id"])]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
ids=sorted(dur)[-200:]
keep=set(ids)
agg=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if int(r["Dispatch_Id"]) in keep: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
ns=sum(dur[i] for i in ids)
gui=agg["GRBM_GUI_ACTIVE"]/8
print("persist=$P  launches %d  avg %.4f ms  clock %.3f GHz  mfma busy %.3f of SIMD-cycles  cycles per launch %.0f k" % (len(ids), ns/len(ids)/1e6, gui/ns, agg["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui*1024), gui/len(ids)/1e3))
PY
  find $R/gpurun_out/clk_p$P -name "*.csv" -size +2M -delete
done
