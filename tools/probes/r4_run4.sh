cd /root/repo
export VIRNET_WX4_MIN_WGS=0 VIRNET_WX4_MIN_TILES=0 VIRNET_WX4_MIN_FILL=0
for m in pre res; do
for f in f16x3 wx4; do
  if [ $f = wx4 ]; then AB="--ab VIRNET_WX4_ROWS=16,8"; else AB=""; fi
  echo "== form $f mode $m"
  VIRNET_CONV_FORM=$f timeout 300 python tools/bench_conv.py --shapes q0,q1,q2,b4,b4_1,b4_2,one,one1,one2,r0,r1,r2 --mode $m --iters 30 $AB 2>&1 | grep -v amdgpu.ids
done; done
