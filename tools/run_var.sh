for v in "1 0" "2 0" "1 1" "2 1"; do set -- $v; echo "== MREP=$1 NREP1=$2";
 if [ "$2" = 1 ]; then export VIRNET_FORCE_NREP=1; else unset VIRNET_FORCE_NREP; fi
 VIRNET_FORCE_MREP=$1 python tools/bench_conv.py --mode pre --shapes one,one1,one2,q0,q1,q2,b4,b4_1,b4_2 2>&1 | grep median | awk '{print $2, $5, $10}'
done
