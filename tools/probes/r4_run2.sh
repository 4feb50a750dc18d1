cd /root/repo
export VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_timing.so
for m in pre res; do for s in l0 l1; do echo "== $s $m rows 8"; VIRNET_WX4_ROWS=8 timeout 300 python tools/wx4h_timeline.py --shape $s --mode $m 2>&1 | grep -v amdgpu.ids; done; done
echo "== l0 pre rows 16"; VIRNET_WX4_ROWS=16 timeout 300 python tools/wx4_timeline.py --shape l0 --mode pre 2>&1 | grep -v amdgpu.ids | head -6
