cd /root/repo
for v in 1000000000 192 128 96; do
echo "VIRNET_WX4_MIN_SLAB_WGS=$v"
VIRNET_WX4_MIN_SLAB_WGS=$v python tools/bench_latency.py --iters 200 2>&1 | grep -v amdgpu
VIRNET_WX4_MIN_SLAB_WGS=$v python tools/probes/sisr_latency.py 2>&1 | grep "(1, 3" | grep eager
done
