#!/bin/bash
# tools/build_ledger.sh NAME MASK [TILEMOD] -> virnet_amd/lib/libvirnet_hip_led_NAME.so: the shipped objects with conv_f16_wx4.hip rebuilt as
# the energy-ledger probe WX4_LEDGER=MASK (see the stage lambda of that file).  Tuning builds; never shipped.
set -e
NAME=$1; MASK=$2; TM=${3:-4}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/build/ledger; mkdir -p $OBJ
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -DWX4_LEDGER=$MASK -DWX4_LEDGER_TILEMOD=$TM \
  -c $ROOT/virnet_amd/csrc/conv_f16_wx4.hip -o $OBJ/wx4_$NAME.o
OTHERS=$(ls $ROOT/build/csrc/*.o | grep -v conv_f16_wx4.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/virnet_amd/lib/libvirnet_hip_led_$NAME.so $OBJ/wx4_$NAME.o $OTHERS
echo built led_$NAME mask=$MASK tilemod=$TM
