#!/bin/bash
# Developer tool: scripts/build_variant.sh NAME FILE.hip "-DFLAG ..."  builds metricsfm_amd/libmsfm_NAME.so in which FILE.hip is
# compiled with the extra flags and every other object is the default build's (for A/B timing through MSFM_LIB, scripts/knn_ab.py).
set -e
cd "$(dirname "$0")/../metricsfm_amd/csrc"
name=$1; src=$2; flags=$3
make -s -j4
mkdir -p /tmp/msfm_variants
obj=/tmp/msfm_variants/${name}_${src%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c $src -o $obj
objs=""
for f in ctx ba chol knn tri geo tracks pose chain multi; do
  if [ "$f.hip" == "$src" ]; then objs="$objs $obj"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmsfm_${name}.so $objs
echo built metricsfm_amd/libmsfm_${name}.so
