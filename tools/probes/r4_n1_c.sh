cd /root/repo
python -m pytest tests/test_guard_gpu.py tests/test_e2e_gpu.py tests/test_host.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_latency.py --iters 200
python tools/bench_latency.py --iters 200 --graph
python tools/probes/sisr_latency.py 2>&1 | grep -v amdgpu
