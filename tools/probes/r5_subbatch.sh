#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python tools/probes/subbatch_chain.py --sb 32,16,8,6,4,3,2,1 > $O/subbatch_l0.log 2>&1
VIRNET_WX4_ROWS=16 python tools/probes/subbatch_chain.py --sb 32,8,4,2 > $O/subbatch_l0_rows16.log 2>&1
python tools/probes/subbatch_chain.py --shape 32,128,128,192 --sb 32,16,8,4 > $O/subbatch_l1.log 2>&1
python tools/probes/subbatch_chain.py --shape 32,64,64,288 --sb 32,16,8 > $O/subbatch_l2.log 2>&1
cat $O/subbatch_l0.log $O/subbatch_l0_rows16.log $O/subbatch_l1.log $O/subbatch_l2.log
