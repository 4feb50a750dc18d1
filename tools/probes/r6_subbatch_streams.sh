#!/bin/bash
# round 6: the level's res-block chain in sub-batches on several streams (Infinity-Cache residency without the launch-boundary loss)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
python tools/probes/subbatch_chain.py --sb 32,4,2,1 --streams 1,2,3,4 > $O/subbatch_streams_l0.log 2>&1
VIRNET_WX4_ROWS=16 python tools/probes/subbatch_chain.py --sb 32,4,2,1 --streams 1,2,3,4 > $O/subbatch_streams_l0_rows16.log 2>&1
python tools/probes/subbatch_chain.py --shape 32,128,128,192 --sb 32,8,4,2 --streams 1,2,3,4 > $O/subbatch_streams_l1.log 2>&1
VIRNET_WX4_ROWS=16 python tools/probes/subbatch_chain.py --shape 32,128,128,192 --sb 32,8,4,2 --streams 1,2,3,4 > $O/subbatch_streams_l1_rows16.log 2>&1
python tools/probes/subbatch_chain.py --shape 32,64,64,288 --sb 32,16,8,4 --streams 1,2,3 > $O/subbatch_streams_l2.log 2>&1
tail -n 20 $O/subbatch_streams_l0.log $O/subbatch_streams_l0_rows16.log $O/subbatch_streams_l1.log $O/subbatch_streams_l1_rows16.log $O/subbatch_streams_l2.log
python -m pytest tests/test_autograph_gpu.py tests/test_guard_gpu.py tests/test_e2e_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/tests_a.log
cat $O/tests_a.log
