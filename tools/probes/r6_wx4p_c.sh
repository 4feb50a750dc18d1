#!/bin/bash
# round 6: persistent form, workgroups per XCD swept (512 = one item per workgroup: the new code without persistence)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
rm -f $O/wx4p_ledger_c.log
for W in 32 64 128 512 0; do
  if [ $W = 0 ]; then E="VIRNET_WX4_PERSIST=0"; else E="VIRNET_WX4_PERSIST=1,VIRNET_WX4_PERSIST_WGS=$W"; fi
  python tools/probes/joule_ledger.py --sweep shipped --shapes l0 --modes pre,res --seconds 2 --env $E 2>&1 | grep -v "^ROWS\|amdgpu.ids" | tail -2 | sed "s/^shipped/wgs=$W/" >> $O/wx4p_ledger_c.log
done
cat $O/wx4p_ledger_c.log
