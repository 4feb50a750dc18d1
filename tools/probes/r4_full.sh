cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4_gpu_tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_a.json 2> gpurun_out/r4_bench_a.err
tail -3 gpurun_out/r4_gpu_tests.log; cat gpurun_out/r4_bench_a.json | cut -c1-1500
