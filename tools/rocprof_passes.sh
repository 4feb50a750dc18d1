#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats            -> per-kernel durations (must agree with bench.py's HIP-event numbers)
#   2. --pmc SQ_* / GRBM (own pass)      -> MFMA busy, wave cycles, stalls
#   3. --pmc FETCH_SIZE, 4. --pmc WRITE_SIZE (own passes; TCC slots do not fit both) -> HBM bytes per launch
# Counters are never combined with sys/hip/hsa tracing (node-crash guard of this pool).
# Usage: tools/rocprof_passes.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-configs $*"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- $BENCH --steps 5 --warmup 2 > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
  --kernel-trace -d "$OUT/pmc_sq" -o p --output-format csv -- $BENCH --steps 1 --warmup 1 --no-roofline > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU \
  --kernel-trace -d "$OUT/pmc_sq2" -o p --output-format csv -- $BENCH --steps 1 --warmup 1 --no-roofline > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o p --output-format csv -- $BENCH --steps 1 --warmup 1 --no-roofline > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o p --output-format csv -- $BENCH --steps 1 --warmup 1 --no-roofline > "$OUT/pmc_write.log" 2>&1
python $ROOT/tools/summarize_r06.py "$OUT" "$OUT" $TAG > "$OUT/summary.txt" 2> "$OUT/summary.err"
tail -3 "$OUT/trace.log"; tail -5 "$OUT/summary.err"; cat "$OUT/summary.txt"; cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"; find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
# keep gpurun_out small: drop the raw per-dispatch CSVs except the stats
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
