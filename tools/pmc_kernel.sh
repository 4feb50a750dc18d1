# PMC snapshot of one kernel: tools/pmc_kernel.sh <kernel substring> <python script + args...>   (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmck -o p --output-format csv -- python $R/$@ > /dev/null 2>&1
python - <<PY
import csv,glob,collections
d="$R/gpurun_out/pmck"
f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)[0]
t=glob.glob(d+"/**/*kernel_trace.csv",recursive=True)[0]
dur={}
for r in csv.DictReader(open(t)):
    if "$K" in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
agg=collections.defaultdict(float); n=set()
for r in csv.DictReader(open(f)):
    if "$K" in r["Kernel_Name"]:
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
ns=sum(dur[i] for i in n); gui=agg["GRBM_GUI_ACTIVE"]/8
wc=agg["SQ_WAVE_CYCLES"]
print("$K launches %d avg %.3f ms clock %.3f GHz | mfma busy %.1f%% of SIMD-cycles | resident waves/SIMD %.2f | of wave-cycles: wait_any %.1f%% wait_inst %.1f%% active %.1f%%" % (len(n), ns/len(n)/1e6, gui/ns, 100*agg["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui*1024), wc*4/(gui*1024), 100*agg["SQ_WAIT_ANY"]/wc, 100*agg["SQ_WAIT_INST_ANY"]/wc, 100*agg["SQ_ACTIVE_INST_ANY"]/wc))
PY
