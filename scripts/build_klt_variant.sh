#!/bin/bash
# build a variant of libov2hip.so whose klt.hip is compiled with extra flags: scripts/build_klt_variant.sh <name> <flags...>
# (objects of the other sources come from build/obj, i.e. run __graft_entry__.build() first)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/ov2slam_amd/lib/variants $R/build/variants
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -Wno-unused-const-variable "$@" \
  -I $R/include -c -o $R/build/variants/klt_$name.o $R/ov2slam_amd/csrc/klt.hip
objs=$(ls $R/build/obj/*.o | grep -v klt.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ov2slam_amd/lib/variants/$name.so $objs $R/build/variants/klt_$name.o
echo built $R/ov2slam_amd/lib/variants/$name.so
