#!/bin/bash
# round 6: the persistent form after the register work -- parity, timeline, time / joules against the per-item form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_wx4p_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/wx4p_tests.log
cat $O/wx4p_tests.log
( export VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_timing.so
  for s in l0 l1 l2; do for m in pre res; do python tools/wx4p_timeline.py --shape $s --mode $m; done; done
  VIRNET_WX4_PERSIST=0 python tools/wx4_timeline.py --shape l0 --mode pre | head -3 ) 2>&1 | grep -v amdgpu.ids > $O/wx4p_timeline_b.log
cat $O/wx4p_timeline_b.log
for P in 0 1 0 1; do
  VIRNET_WX4_PERSIST=$P python tools/probes/joule_ledger.py --sweep shipped --shapes l0,l1,l2 --modes pre,res --seconds 2 --tag persist$P 2>&1 | grep -v "^ROWS\|amdgpu.ids" | tail -6 | sed "s/^shipped/persist=$P/" >> $O/wx4p_ledger_b.log
done
cat $O/wx4p_ledger_b.log
