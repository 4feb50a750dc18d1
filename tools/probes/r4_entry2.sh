cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "entry" 2>&1 | tail -2
for rep in 1 2; do for v in 0 1; do
VIRNET_ENTRY_FUSED=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused=$v bench', d['value'], d['ms_per_step'], d['roofline']['by_kernel_ms_per_step'].get('conv_f16<cout=96>'), d['roofline']['by_kernel_ms_per_step'].get('conv_f16<cout=64>'))"
done; done
