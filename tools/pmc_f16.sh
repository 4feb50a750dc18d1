# PMC snapshot of conv_f16_kernel on one bench_conv shape (run through gpurun from the repo root): tools/pmc_f16.sh l0 pre
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SH=${1:-l0}; MODE=${2:-pre}
rm -rf $R/gpurun_out/pmcf; mkdir -p $R/gpurun_out/pmcf
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmcf/a -o p --output-format csv -- python $R/tools/bench_conv.py --shapes $SH --mode $MODE --iters 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace -d $R/gpurun_out/pmcf/b -o p --output-format csv -- python $R/tools/bench_conv.py --shapes $SH --mode $MODE --iters 5 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcf/c -o p --output-format csv -- python $R/tools/bench_conv.py --shapes $SH --mode $MODE --iters 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmcf/d -o p --output-format csv -- python $R/tools/bench_conv.py --shapes $SH --mode $MODE --iters 5 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
K="${KERNEL:-conv_f16_kernel}"
out={}
for sub in "abcd":
    d="$R/gpurun_out/pmcf/"+sub
    fs=glob.glob(d+"/**/*counter_collection.csv",recursive=True); ts=glob.glob(d+"/**/*kernel_trace.csv",recursive=True)
    if not fs or not ts: print("pass",sub,"missing"); continue
    dur={}
    for r in csv.DictReader(open(ts[0])):
        if K in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    agg=collections.defaultdict(float); n=set()
    for r in csv.DictReader(open(fs[0])):
        if K in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    L=max(len(n),1)
    for k,v in agg.items(): out[k]=v/L
    out["ms_"+sub]=sum(dur[i] for i in n)/L/1e6
gui=out.get("GRBM_GUI_ACTIVE",0)/8; wc=out.get("SQ_WAVE_CYCLES",1)
ns=out.get("ms_a",0)*1e6
print("per launch: %.3f ms, clock %.3f GHz, mfma busy %.1f%% of SIMD-cycles, resident waves/SIMD %.2f; of wave-cycles: wait_any %.1f%% wait_inst %.1f%% active %.1f%%" % (out.get("ms_a",0), gui/max(ns,1), 100*out.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/max(gui*1024,1), wc*4/max(gui*1024,1), 100*out.get("SQ_WAIT_ANY",0)/wc, 100*out.get("SQ_WAIT_INST_ANY",0)/wc, 100*out.get("SQ_ACTIVE_INST_ANY",0)/wc))
for k in ("SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_LDS","SQ_WAIT_INST_LDS","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_VMEM_RD","SQ_INSTS_VALU_MFMA_MOPS_F16","FETCH_SIZE","WRITE_SIZE","SQ_WAVES"):
    print("  %-30s %.4g" % (k, out.get(k,float("nan"))))
PY
