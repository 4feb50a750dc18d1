cd /root/repo
python -m pytest tests/test_t_emit_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --task train --steps 10 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
VIRNET_T_EMIT=0 run "no-emit"
VIRNET_T_EMIT=0 VIRNET_WX4_ROWS=8 run "no-emit rows8"
VIRNET_WX4_EMIT_ROWS=16 run "emit16"
VIRNET_WX4_EMIT_ROWS=8 run "emit8"
VIRNET_WX4_EMIT_ROWS=8 VIRNET_WX4_ROWS=8 run "emit8 all-rows8"
done
