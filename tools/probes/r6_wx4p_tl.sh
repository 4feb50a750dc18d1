#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
export VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_timing.so
for m in pre; do
python tools/wx4p_timeline.py --shape l0 --mode $m --loaded 400
VIRNET_WX4_PERSIST=0 python tools/wx4_timeline.py --shape l0 --mode $m --loaded 400 | head -4
python tools/wx4p_timeline.py --shape l0 --mode $m
VIRNET_WX4_PERSIST=0 python tools/wx4_timeline.py --shape l0 --mode $m | head -4
done 2>&1 | grep -v amdgpu.ids | tee $O/wx4p_timeline_d.log
