#!/bin/bash
# A/B of the wx4 kernel on the three res-block shapes (conv1-type "pre", conv2-type "res") + the workgroup timeline of a timing build
cd /root/repo
export VIRNET_CONV_FORM=wx4
for rep in 1 2; do
for v in "$@"; do
  for m in pre res; do
    VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip$v.so python tools/bench_conv.py --shapes l0,l1,l2 --mode $m --iters 30 2>&1 | grep "^lib"
  done
done
done
if [ -f virnet_amd/lib/libvirnet_hip_timing.so ]; then
for m in pre res; do echo "== timeline $m"; VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_timing.so python tools/wx4_timeline.py --shape l0 --mode $m 2>&1 | grep -v amdgpu.ids | head -12; done
fi
