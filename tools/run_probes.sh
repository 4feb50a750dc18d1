#!/bin/bash
# tools/run_probes.sh SHAPES NAME... : bench_conv for the shipped library and each variant build
SH=$1; shift
python tools/bench_conv.py --mode pre --shapes $SH
for v in "$@"; do VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_$v.so python tools/bench_conv.py --mode pre --shapes $SH; done
