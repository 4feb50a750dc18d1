cd /root/repo
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "exit or thin" 2>&1 | tail -8
timeout 300 python tools/probes/exit_time.py 2>&1 | grep form
