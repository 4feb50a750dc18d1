cd /root/repo
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "bf16" 2>&1 | tail -3
run() { python bench.py --task train --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do run "bf16"; done
