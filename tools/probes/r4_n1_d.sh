cd /root/repo
export VIRNET_WX4_MIN_WGS=1
for r in 8 16; do
export VIRNET_WX4_ROWS=$r
echo "rows $r"
python tools/bench_conv.py --shapes q0,q1,q2,r1 --mode pre --iters 30 --ab VIRNET_WX4_NREP=3,2,1 2>&1 | grep -v amdgpu
done
unset VIRNET_WX4_ROWS
export VIRNET_WX4_MIN_WGS=100000
echo direct
python tools/bench_conv.py --shapes q0,q1,q2,r1 --mode pre --iters 30 2>&1 | grep -v amdgpu
