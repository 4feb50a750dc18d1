cd /root/repo
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -22 > gpurun_out/r4_gpu_tests.log
echo "suite wall $(( $(date +%s) - T0 )) s" >> gpurun_out/r4_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r4_gpu_tests.log
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_b.json 2> gpurun_out/r4_bench_b.err
echo "bench wall $(( $(date +%s) - T0 )) s" >> gpurun_out/r4_gpu_tests.log
python - <<PY >> gpurun_out/r4_gpu_tests.log
import json
d=json.load(open("gpurun_out/r4_bench_b.json"))
print("bench", d["value"], d["ms_per_step"], d["steady_state"]["value"], d["steady_state"]["power"]["socket_w_mean"])
for k,v in d["configs"].items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("error"))
print(d["roofline"]["by_kernel_ms_per_step"])
PY
tail -40 gpurun_out/r4_gpu_tests.log
