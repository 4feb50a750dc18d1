#!/bin/bash
# round 6, last pass from HEAD: GPU suite, smoke, default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_final2; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.log 2> $O/bench_default.err; head -c 700 $O/bench_default.log; echo
