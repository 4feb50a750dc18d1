# cache-path counters of the exit kernel (is the 4-touches-per-line pattern served by L1 or by L2?): through gpurun
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/exitpmc
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCP|TCC|TA|TD)_[A-Z0-9_a-z]+" | sort -u > $R/gpurun_out/exitpmc/avail.txt
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/exitpmc/p$i -o p --output-format csv -- python $R/tools/probes/exit_time.py > $R/gpurun_out/exitpmc/run$i.log 2>&1
done
python - <<PY
import csv,glob,collections
R="$R/gpurun_out/exitpmc"
for d in sorted(glob.glob(R+"/p*")):
    fs=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not fs: print(d,"no counters"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        if "conv_exit" in k:
            key=k.split("(")[0][-24:]+" grid "+r.get("Grid_Size","?")
            agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[key].add(r["Dispatch_Id"])
    for key in agg:
        print(key, {c: "%.4g"%(v/len(n[key])) for c,v in agg[key].items()}, "launches",len(n[key]))
PY
