cd /root/repo
timeout 900 python -m pytest tests/test_guard_gpu.py tests/test_conv_wx4_gpu.py tests/test_e2e_gpu.py tests/test_backward_gpu.py -x -q -m gpu 2>&1 | tail -30
