cd /root/repo
export VIRNET_CONV_FORM=wx4 VIRNET_WX4_ROWS=16 VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_probe2x.so
for s in l0 l1 l2; do
python tools/bench_conv.py --shapes $s --mode pre --iters 30 --ab WX4_PROBE_REPS=1,2,3 2>&1 | grep -v amdgpu.ids
done
