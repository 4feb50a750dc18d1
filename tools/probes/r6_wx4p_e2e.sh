#!/bin/bash
# round 6: the persistent form end to end (the metric's step), interleaved A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
rm -f $O/wx4p_e2e.log
for i in 1 2 3; do for P in 0 1; do
  VIRNET_WX4_PERSIST=$P python bench.py --no-cpu-baseline --no-configs --steps 40 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('persist=$P', d['value'], d['ms_per_step'], d['power']['socket_w_mean'], {k:v for k,v in r['by_kernel_ms_per_step'].items() if 'wx4' in k})" >> $O/wx4p_e2e.log
done; done
cat $O/wx4p_e2e.log
