#!/bin/bash
# tools/mkvariant.sh NAME TU "FLAGS": tools/scratch/libaisx_NAME.so = the EXPERIMENTS build of the library (lib/libaisx_exp.so:
# AISX_* knobs, alternative kernels) with translation unit TU (aisx_lib | aisx_msk | aisx_stages | aisx_chain) rebuilt with
# extra compiler FLAGS (for tools/ab_bench.py, tools/native/corrbench)
set -e
cd "$(dirname "$0")/../gr-ais_amd"
make -s
name=$1; tu=$2; flags=$3
if [ "$tu" = aisx_stages ]; then flags="-fno-slp-vectorize $flags"; fi  # (as gr-ais_amd/Makefile builds that unit)
if [ "$tu" = aisx_lib ]; then flags="-Xclang -target-feature -Xclang -load-store-opt $flags"; fi
mkdir -p ../tools/scratch/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -DAISX_EXPERIMENTS $flags -c -o ../tools/scratch/obj/${tu}_${name}.o csrc/${tu}.hip 2> >(grep -v "is not a recognized feature for this target" >&2)
objs=""
for o in aisx_lib aisx_msk aisx_stages aisx_chain; do
  if [ $o = $tu ]; then objs="$objs ../tools/scratch/obj/${tu}_${name}.o"; else objs="$objs build_exp/$o.hip.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/scratch/libaisx_${name}.so $objs build_exp/aisx_framing.cpp.o
echo tools/scratch/libaisx_${name}.so
