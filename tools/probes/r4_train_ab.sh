# training-step bench lines of HEAD (fp32-class, bf16 as configs[4] writes it, SISR step) on one box
cd /root/repo
python bench.py --task train --steps 10 --warmup 3 --no-cpu-baseline --no-configs > gpurun_out/tb_f32.json 2> gpurun_out/tb_f32.err
python bench.py --task train --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-configs > gpurun_out/tb_bf16.json 2> gpurun_out/tb_bf16.err
python bench.py --task train_sisr --steps 10 --warmup 3 --no-cpu-baseline --no-configs > gpurun_out/tb_sisr.json 2> gpurun_out/tb_sisr.err
for f in f32 bf16 sisr; do tail -c 400 gpurun_out/tb_$f.err; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/tb_$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["ms_per_step"], d.get("power"))
except Exception as e:
    print("$f", "failed", e)
PY
done
