#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags] -> virnet_amd/lib/libvirnet_hip_NAME.so (tuning builds; not shipped)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/build/variant_$NAME
mkdir -p $OBJ
for f in $(cd $ROOT/virnet_amd/csrc && ls *.hip *.cpp); do
  rm -f $OBJ/${f%.*}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip "$@" -c $ROOT/virnet_amd/csrc/$f -o $OBJ/${f%.*}.o &
done
wait
for f in $(cd $ROOT/virnet_amd/csrc && ls *.hip *.cpp); do [ -f $OBJ/${f%.*}.o ] || { echo "FAILED: $f"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/virnet_amd/lib/libvirnet_hip_$NAME.so $OBJ/*.o
echo built $NAME
