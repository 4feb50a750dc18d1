cd /root/repo
run() { python bench.py --task train $2 --steps 10 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
VIRNET_BIAS_FUSED=0 run "separate" "--dtype bf16"; run "fused" "--dtype bf16"
done
VIRNET_BIAS_FUSED=0 run "separate" ""; run "fused" ""
