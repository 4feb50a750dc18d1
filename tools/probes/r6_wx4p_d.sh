#!/bin/bash
# round 6, second pass on the persistent form (zero-C first products): parity, light timeline, launch time against the per-item form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_wx4p_gpu.py -x -q -m gpu 2>&1 | tail -2
( export VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_timing_light.so
  for m in pre res; do python tools/wx4p_timeline.py --shape l0 --mode $m --loaded 400 2>/dev/null | grep "per item\|workgroup span"; done ) | tee $O/wx4p_timeline_light2.log
rm -f $O/wx4p_ledger_d.log
for P in 0 1 0 1; do
  VIRNET_WX4_PERSIST=$P python tools/probes/joule_ledger.py --sweep shipped --shapes l0,l1 --modes pre,res --seconds 2 2>&1 | grep -v "^ROWS\|amdgpu.ids" | tail -4 | sed "s/^shipped/persist=$P/" >> $O/wx4p_ledger_d.log
done
cat $O/wx4p_ledger_d.log
