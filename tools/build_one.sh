#!/bin/bash
# tools/build_one.sh NAME UNIT [hipcc flags] -> virnet_amd/lib/libvirnet_hip_NAME.so: the shipped objects (build/csrc) with ONE unit
# (virnet_amd/csrc/UNIT.hip) rebuilt with extra flags.  Tuning / probe builds; never shipped.  Select with VIRNET_HIP_LIB.
set -e
NAME=$1; UNIT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/build/one; mkdir -p $OBJ
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip "$@" -c $ROOT/virnet_amd/csrc/$UNIT.hip -o $OBJ/${UNIT}_$NAME.o
OTHERS=$(ls $ROOT/build/csrc/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/virnet_amd/lib/libvirnet_hip_$NAME.so $OBJ/${UNIT}_$NAME.o $OTHERS
echo built $NAME
