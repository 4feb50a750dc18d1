#!/bin/bash
# round 5, final measurements from HEAD: GPU suite, default bench line, latency table, rocprofv3 passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_final; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.log 2> $O/bench_default.err; tail -c 1500 $O/bench_default.log
python tools/bench_latency.py --iters 200 > $O/latency.txt 2>&1; python tools/probes/sisr_latency.py >> $O/latency.txt 2>&1; grep -v amdgpu $O/latency.txt
bash tools/rocprof_passes.sh r05 > $O/rocprof.log 2>&1; tail -20 $O/rocprof.log
