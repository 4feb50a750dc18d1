cd /root/repo
timeout 1200 python -m pytest tests/test_sisr_train_gpu.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -30
