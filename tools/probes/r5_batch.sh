#!/bin/bash
# round 5: does the forward run faster in sub-batches whose tensors stay in the Infinity Cache? (ledger: HBM+fabric = 21 % of the l0 launch's energy)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python -m pytest tests/test_guard_gpu.py -x -q > $O/guard_tests.log 2>&1; tail -3 $O/guard_tests.log
for b in 1 2 3 4 6 8 12 16 32; do
python bench.py --batch $b --steps $((640 / b > 200 ? 200 : 640 / b)) --warmup 5 --no-cpu-baseline --no-configs --no-roofline 2>$O/batch_err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b', d['value'], d['ms_per_step'], (d.get('power') or {}).get('socket_w_mean'), (d.get('power') or {}).get('sclk_mhz_mean'))"
done 2>&1 | tee $O/batch_sweep.log
