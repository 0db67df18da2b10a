#!/bin/bash
# tools/mkvariant.sh NAME TU "FLAGS": tools/scratch/libaisx_NAME.so = the product library with translation
# unit TU (aisx_lib | aisx_msk | aisx_stages | aisx_chain) rebuilt with extra compiler FLAGS (for tools/ab_bench.py)
set -e
cd "$(dirname "$0")/../gr-ais_amd"
make -s
name=$1; tu=$2; flags=$3
if [ "$tu" = aisx_stages ]; then flags="-fno-slp-vectorize $flags"; fi  # (as gr-ais_amd/Makefile builds that unit)
mkdir -p ../tools/scratch/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $flags -c -o ../tools/scratch/obj/${tu}_${name}.o csrc/${tu}.hip
objs=""
for o in aisx_lib aisx_msk aisx_stages aisx_chain; do
  if [ $o = $tu ]; then objs="$objs ../tools/scratch/obj/${tu}_${name}.o"; else objs="$objs build/$o.hip.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/scratch/libaisx_${name}.so $objs build/aisx_framing.cpp.o
echo tools/scratch/libaisx_${name}.so
