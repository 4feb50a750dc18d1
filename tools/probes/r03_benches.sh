#!/bin/bash
# the round's bench lines on one box: headline shape, configs[1], SISR, training steps, + the 64-channel wx4 experiment
cd /root/repo
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_256.json 2> gpurun_out/r03_bench_256.err
VIRNET_WX4_MIN_COUT=64 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench_256_wx64.json 2>/dev/null
python bench.py --size 128 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench_128.json 2>/dev/null
python bench.py --task sisr --steps 20 --warmup 5 > gpurun_out/r03_bench_sisr.json 2>/dev/null
python bench.py --task train --steps 10 --warmup 3 > gpurun_out/r03_bench_train.json 2>/dev/null
python bench.py --task train --dtype bf16 --steps 10 --warmup 3 > gpurun_out/r03_bench_train_bf16.json 2>/dev/null
python bench.py --task train_sisr --steps 10 --warmup 3 > gpurun_out/r03_bench_train_sisr.json 2>/dev/null
for f in 256 256_wx64 128 sisr train train_bf16 train_sisr; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r03_bench_$f.json")); print("$f", d["value"], d["ms_per_step"], d.get("roofline",{}).get("by_kernel_ms_per_step"))
except Exception as e: print("$f FAILED", e)
PY
done
{ echo "# wx4 (default)"; python tools/bench_latency.py; python tools/bench_latency.py --graph; echo "# VIRNET_CONV_FORM=f16x3 (round 2's kernels)"; VIRNET_CONV_FORM=f16x3 python tools/bench_latency.py; } 2>/dev/null | grep -v amdgpu > gpurun_out/r03_latency.txt
cat gpurun_out/r03_latency.txt
