#!/bin/bash
# round 6: first run of the persistent form -- parity, then time / joules against the per-item form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_wx4p_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/wx4p_tests.log
cat $O/wx4p_tests.log
for P in 0 1; do
  VIRNET_WX4_PERSIST=$P python tools/probes/joule_ledger.py --sweep shipped --shapes l0,l1,l2 --modes pre,res --seconds 2 --tag persist$P 2>&1 | grep -v "^ROWS" | tail -6 > $O/wx4p_ledger_p$P.log
  cat $O/wx4p_ledger_p$P.log
done
