#!/bin/bash
# round 6, measurements from HEAD: GPU suite, default bench line, latency table in every call mode, rocprofv3 passes (denoise step), kernel stats
# of the SISR forward and of the two training steps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_final; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.log 2> $O/bench_default.err; head -c 1200 $O/bench_default.log; echo
python tools/bench_latency.py --all-modes --iters 200 2>&1 | grep -v amdgpu > $O/latency_modes.txt; cat $O/latency_modes.txt
bash tools/rocprof_passes.sh r06 > $O/rocprof.log 2>&1; tail -20 $O/rocprof.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for T in sisr train train_bf16; do
  A="--task $T"; [ $T = train_bf16 ] && A="--task train --dtype bf16"
  rm -rf $R/gpurun_out/prof_r06_$T; mkdir -p $R/gpurun_out/prof_r06_$T
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06_$T -o t --output-format csv -- python $R/bench.py $A --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $R/gpurun_out/prof_r06_$T/log.txt 2>&1
  f=$(find $R/gpurun_out/prof_r06_$T -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/r6_final/r06_${T}_kernel_stats.csv
  find $R/gpurun_out/prof_r06_$T -name "*kernel_trace.csv" -size +4M -delete
  head -6 $f | cut -c1-160
done
