cd /root/repo
run() { python bench.py --task train $2 --steps 10 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for mode in "" "--dtype bf16"; do
VIRNET_T_EMIT=0 run "no-emit" "$mode"
run "emit+qt" "$mode"
VIRNET_HIP_LIB=/root/repo/virnet_amd/lib/libvirnet_hip_noqt.so run "emit-noqt" "$mode"
done
done
