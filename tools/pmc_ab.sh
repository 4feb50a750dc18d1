cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in both now; do
  export VIRNET_HIP_LIB=$R/virnet_amd/lib/libvirnet_hip_$v.so
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --kernel-trace -d $R/gpurun_out/ab_$v/a -o p --output-format csv -- python $R/tools/bench_conv.py --shapes l1 --iters 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace -d $R/gpurun_out/ab_$v/b -o p --output-format csv -- python $R/tools/bench_conv.py --shapes l1 --iters 3 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
for sub in "ab":
    f=glob.glob("$R/gpurun_out/ab_$v/"+sub+"/**/*counter_collection.csv",recursive=True)
    agg=collections.defaultdict(float); n=set()
    for r in csv.DictReader(open(f[0])):
        if "conv_mfma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    print("$v",sub,len(n),{k:round(v/len(n)/1e6,2) for k,v in sorted(agg.items())})
PY
done
