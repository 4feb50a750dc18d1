cd /root/repo
timeout 1500 python -m pytest tests/test_fullset_gpu.py tests/test_trained_like_gpu.py -x -q -m gpu -s 2>&1 | tail -25
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "eight_rank" 2>&1 | tail -8
