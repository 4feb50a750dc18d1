cd /root/repo
export VIRNET_CONV_FORM=wx4
rocm-smi --showmaxpower --showpower 2>&1 | grep -v "^$" | head -12
for z in 0 1; do for r in 16 8; do BENCH_ZEROS=$z VIRNET_WX4_ROWS=$r timeout 120 python tools/probes/power_probe.py --shape l0 --mode pre --seconds 3 2>&1 | grep -v amdgpu.ids; done; done
VIRNET_WX4_ROWS=16 timeout 120 python tools/probes/power_probe.py --shape l2 --mode pre --seconds 3 2>&1 | grep -v amdgpu.ids
