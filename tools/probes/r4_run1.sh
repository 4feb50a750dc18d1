cd /root/repo
export VIRNET_CONV_FORM=wx4
timeout 600 python -m pytest tests/test_conv_wx4_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r4_wx4h_tests.log
for m in pre res; do
  timeout 300 python tools/bench_conv.py --shapes l0,l1,l2,s64 --mode $m --iters 20 --ab VIRNET_WX4_ROWS=16,8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4_wx4h_ab.log
done
cat gpurun_out/r4_wx4h_tests.log gpurun_out/r4_wx4h_ab.log
