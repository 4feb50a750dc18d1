cd /root/repo
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 2>&1 | tail -45 > gpurun_out/r4_gpu_tests.log
echo "suite wall $(( $(date +%s) - T0 )) s" >> gpurun_out/r4_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> gpurun_out/r4_gpu_tests.log
python bench.py --task sisr --steps 20 --warmup 5 --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sisr', d['value'], d['ms_per_step'], d['roofline']['by_kernel_ms_per_step'])" >> gpurun_out/r4_gpu_tests.log
tail -50 gpurun_out/r4_gpu_tests.log
