#!/bin/bash
# round 5: joule ledger of the dominant launch -- every probe build, l0 conv1/conv2-type and l1, then the same at a pinned clock
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
V=shipped,base,nomfma,norda,nordb,nodma,nostage,novst,nopix,l2,mall,noepi,mfmaonly
python tools/probes/joule_ledger.py --sweep $V --shapes l0 --modes pre,res --seconds 3 > $O/ledger_l0.log 2>&1
python tools/probes/joule_ledger.py --sweep base,nomfma,norda,nodma,nostage,l2,noepi,mfmaonly --shapes l1,l2 --modes pre --seconds 3 > $O/ledger_l12.log 2>&1
BENCH_ZEROS=1 python tools/probes/joule_ledger.py --sweep base,nomfma,mfmaonly,l2,noepi --shapes l0 --modes pre --seconds 3 > $O/ledger_l0_zeros.log 2>&1
rocm-smi --setperfdeterminism 1500 > $O/perfdet.log 2>&1
python tools/probes/joule_ledger.py --sweep $V --shapes l0 --modes pre --seconds 3 > $O/ledger_l0_1500.log 2>&1
rocm-smi --resetperfdeterminism >> $O/perfdet.log 2>&1
tail -n 3 $O/ledger_l0.log
