# effective shader clock per build: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (tuning aid)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = default ]; then unset VIRNET_HIP_LIB; else export VIRNET_HIP_LIB=$R/virnet_amd/lib/libvirnet_hip_$v.so; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/clk_$v -o p --output-format csv -- python $R/tools/bench_conv.py --shapes ${SHAPES:-l1} --iters 5 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
d="$R/gpurun_out/clk_$v"
f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)[0]
t=glob.glob(d+"/**/*kernel_trace.csv",recursive=True)[0]
dur={}
for r in csv.DictReader(open(t)):
    if "conv_mfma" in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
agg=collections.defaultdict(float); n=set()
for r in csv.DictReader(open(f)):
    if "conv_mfma" in r["Kernel_Name"]:
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
ns=sum(dur[i] for i in n)
gui=agg["GRBM_GUI_ACTIVE"]/8
print("%-8s launches %d avg %.3f ms clock %.3f GHz mfma_busy %.1f%% of SIMD-cycles, resident waves/SIMD %.2f" % ("$v",len(n),ns/len(n)/1e6,gui/ns,100*agg["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui*1024),agg["SQ_WAVE_CYCLES"]*4/(gui*1024)))
PY
done
