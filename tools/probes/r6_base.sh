#!/bin/bash
# round 6: baseline of HEAD at the start of the round -- the default bench line and the shipped kernels' joules per launch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
python bench.py > $O/bench_base.json 2> $O/bench_base.err
python tools/probes/joule_ledger.py --sweep shipped --shapes l0,l1,l2 --modes pre,res --seconds 3 > $O/ledger_base.log 2>&1
tail -n 8 $O/ledger_base.log
head -c 1500 $O/bench_base.json
