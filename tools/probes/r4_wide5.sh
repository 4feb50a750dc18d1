cd /root/repo
timeout 1200 python -m pytest tests/test_conv_wx4_gpu.py tests/test_e2e_gpu.py tests/test_sisr_harness.py tests/test_sisr_train_gpu.py tests/test_guard_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for v in 0 1; do
VIRNET_WX4_WIDE=$v python bench.py --task sisr --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); b=d['roofline']['by_kernel_ms_per_step']; print('wide=$v', d['value'], d['ms_per_step'], b.get('conv_wx4<cout=160>'), b.get('conv_wx4<cout=224>'))"
done; done
for v in 0 1; do VIRNET_WX4_WIDE=$v python tools/bench_conv.py --shapes s1 --mode pre --iters 30 2>&1 | grep -v amdgpu; VIRNET_WX4_WIDE=$v python tools/bench_conv.py --shapes s1 --mode res --iters 30 2>&1 | grep -v amdgpu; done
