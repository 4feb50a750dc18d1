cd /root/repo
for m in pre res; do
python tools/bench_conv.py --shapes q2,r2,r1,q1,b4_2 --mode $m --iters 30 --ab VIRNET_F16_SPLIT_WGS=0,128,256,100000
done
VIRNET_F16_SPLIT_WGS=0 python tools/bench_latency.py --iters 100
python tools/bench_latency.py --iters 100
