#!/bin/bash
# tools/build_ledger_s2.sh NAME N -> virnet_amd/lib/libvirnet_hip_led_NAME.so: the shipped objects with conv_f16_s2.hip rebuilt as the probe
# S2_LEDGER=N (1 no MFMA, 2 two of three products, 3 one product).  Tuning builds; never shipped.
set -e
NAME=$1; N=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/build/ledger; mkdir -p $OBJ
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -DS2_LEDGER=$N -c $ROOT/virnet_amd/csrc/conv_f16_s2.hip -o $OBJ/s2_$NAME.o
OTHERS=$(ls $ROOT/build/csrc/*.o | grep -v conv_f16_s2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/virnet_amd/lib/libvirnet_hip_led_$NAME.so $OBJ/s2_$NAME.o $OTHERS
echo built led_$NAME S2_LEDGER=$N
