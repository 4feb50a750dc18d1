#!/bin/bash
# round 6: non-temporal store threshold end to end (128 = default: all three levels of the metric's step; 160: level 2's 151-MB tensors stay cacheable; 512: only level 0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O; rm -f $O/nt_ab.log
for i in 1 2; do for T in 128 160 512; do
  VIRNET_NT_STORE_MB=$T python bench.py --no-cpu-baseline --no-configs --steps 40 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('nt_mb=$T', d['value'], d['ms_per_step'], {k:v for k,v in r['by_kernel_ms_per_step'].items() if 'wx4' in k})" >> $O/nt_ab.log
done; done
cat $O/nt_ab.log
