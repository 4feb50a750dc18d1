# PMC snapshot of the Winograd kernel: tools/pmc_wino.sh <shape> [mode]   (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SH=${1:-l0}; MODE=${2:-pre}
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmcw/a -o p --output-format csv -- python $R/tools/bench_conv.py --shapes $SH --mode $MODE --iters 3 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d $R/gpurun_out/pmcw/b -o p --output-format csv -- python $R/tools/bench_conv.py --shapes $SH --mode $MODE --iters 3 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA --kernel-trace -d $R/gpurun_out/pmcw/c -o p --output-format csv -- python $R/tools/bench_conv.py --shapes $SH --mode $MODE --iters 3 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
for sub in "abc":
    fs=glob.glob("$R/gpurun_out/pmcw/"+sub+"/**/*counter_collection.csv",recursive=True)
    if not fs: print(sub,"no data"); continue
    agg=collections.defaultdict(float); n=set()
    for r in csv.DictReader(open(fs[0])):
        if "conv_wino" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    if not n: print(sub,"no wino dispatch"); continue
    gui=agg["GRBM_GUI_ACTIVE"]/8/len(n)
    print(sub,"launches",len(n),"gui cycles/launch %.0f"%gui)
    for k,v in sorted(agg.items()):
        print("   %-28s per launch %12.0f   per SIMD-cycle %.4f"%(k,v/len(n),v/len(n)/(gui*1024)))
PY
