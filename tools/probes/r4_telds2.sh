cd /root/repo
timeout 900 python -m pytest tests/test_t_emit_gpu.py tests/test_backward_gpu.py -x -q -m gpu 2>&1 | tail -2
run() { python bench.py --task train $2 --steps 10 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do run "f32" ""; run "bf16" "--dtype bf16"; done
