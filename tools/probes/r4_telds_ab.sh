cd /root/repo
timeout 900 python -m pytest tests/test_t_emit_gpu.py tests/test_backward_gpu.py -x -q -m gpu 2>&1 | tail -2
run() { python bench.py --task train --steps 10 --warmup 3 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
VIRNET_HIP_LIB=/root/repo/virnet_amd/lib/libvirnet_hip_tedirect.so run "direct"; run "lds"
done
FORMS=wx4 python tools/probes/emit_conv_ab.py 2>&1 | grep -v amdgpu | head -4
