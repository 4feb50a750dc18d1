cd /root/repo
python -m pytest tests/test_conv_f16_gpu.py tests/test_e2e_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
VIRNET_F16_SPLIT_WGS=0 python tools/probes/sisr_latency.py 2>&1 | tail -2
python tools/probes/sisr_latency.py 2>&1 | tail -2
bash tools/probes/n1_trace.sh 2>&1 | tail -62
