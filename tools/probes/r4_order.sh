cd /root/repo
export VIRNET_CONV_FORM=wx4 VIRNET_WX4_ROWS=16
for rep in 1 2; do for m in pre res; do for v in "" _mpart; do
  echo "== lib$v $m"; VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip$v.so python tools/bench_conv.py --shapes l0,l1,l2 --mode $m --iters 30 2>&1 | grep -E "^default|median"
done; done; done
VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_mpart.so timeout 300 python -m pytest tests/test_conv_wx4_gpu.py -x -q -k "rows16 and (oracle or sweep)" 2>&1 | tail -3
