#!/bin/bash
# Builds a VARIANT of the whole library with extra hipcc flags (A/B runs on one GPU box; never the product):
# tools/build_variant_all.sh NAME "extra flags" -> lerf-pytorch_amd/csrc/build_variants/liblerf_hip_NAME.so (bench.py --lib PATH)
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/lerf-pytorch_amd/csrc
mkdir -p $src/build_variants/$name
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -Wno-unused-function -fno-slp-vectorize -w $extra"
objs=""
for f in lerf_api lerf_kernels lerf_lut_interp lerf_fused lerf_fused_g3 lerf_fused_h32 lerf_fused_h16 lerf_fused_c1 lerf_fused_c4 lerf_metrics lerf_train lerf_transfer lerf_ubench; do
  ( /opt/rocm/bin/hipcc $flags -c -o $src/build_variants/$name/$f.o $src/$f.hip ) &
  objs="$objs $src/build_variants/$name/$f.o"
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o $src/build_variants/liblerf_hip_$name.so $objs
echo "built $src/build_variants/liblerf_hip_$name.so"
