cd /root/repo
for b in 4 8 12 16 24 32 48 64; do
python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-configs --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b', d['value'], d['ms_per_step'], d.get('power',{}).get('socket_w_mean'), d.get('power',{}).get('sclk_mhz_mean'))"
done
