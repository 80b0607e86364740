#!/bin/bash
# dev helper: build a variant of the library next to the default one (A/B with SMG_HIP_LIB=smg-multimodal-grasping_amd/libsmg_<name>.so).
# usage: build_variant.sh name "-DFLAG ..."
name=$1; extra=$2
cd "$(dirname "$0")/../smg-multimodal-grasping_amd/csrc" || exit 1
B=build_$name; mkdir -p $B
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function $extra"
for f in engine forward backward; do /opt/rocm/bin/hipcc $F -c $f.hip -o $B/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/engine.o $B/forward.o $B/backward.o -o ../libsmg_$name.so && echo built ../libsmg_$name.so
