cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "entry" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py tests/test_guard_gpu.py tests/test_eval_extras.py -x -q -m gpu 2>&1 | tail -4
for v in 0 1; do
echo "VIRNET_ENTRY_FUSED=$v"
VIRNET_ENTRY_FUSED=$v python tools/bench_latency.py --iters 200 2>&1 | grep -v amdgpu
VIRNET_ENTRY_FUSED=$v python tools/probes/sisr_latency.py 2>&1 | grep "(1, 3" | grep eager
VIRNET_ENTRY_FUSED=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['by_kernel_ms_per_step'].get('conv_f16<cout=96>'), d['roofline']['by_kernel_ms_per_step'].get('conv_f16<cout=64>'))"
done
