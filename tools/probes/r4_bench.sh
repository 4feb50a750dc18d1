cd /root/repo
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_b.json 2> gpurun_out/r4_bench_b.err
echo "bench wall $(( $(date +%s) - T0 )) s"
grep -E "Error|error|Traceback" gpurun_out/r4_bench_b.err | head
python - <<PY
import json
d=json.load(open("gpurun_out/r4_bench_b.json"))
print(d["value"], d["ms_per_step"], d["power"], d["steady_state"])
print(d["cpu_baseline"])
for k,v in d["configs"].items(): print(k, {kk: v.get(kk) for kk in ("value","ms_per_step","steps","error","power")}, (v.get("roofline") or {}).get("frac_algorithmic"))
print(d["roofline"]["by_kernel_ms_per_step"])
PY
