#!/bin/bash
# Builds a VARIANT of the library in which only csrc/lerf_lut_interp.hip is recompiled with extra flags (A/B and ablation runs of
# the single-pass kernel on one GPU box): tools/build_li_variant.sh NAME "extra hipcc flags"
# -> lerf-pytorch_amd/csrc/build_variants/liblerf_hip_NAME.so (git-ignored; tools/bench_lut_interp.py --lib PATH)
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/lerf-pytorch_amd/csrc
make -s -C $src >/dev/null
mkdir -p $src/build_variants
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -Wno-unused-function -fno-slp-vectorize -w $extra"
/opt/rocm/bin/hipcc $flags -c -o $src/build_variants/li_$name.o $src/lerf_lut_interp.hip
objs=""
for o in lerf_api lerf_kernels lerf_fused lerf_fused_g3 lerf_fused_h32 lerf_fused_h16 lerf_fused_c1 lerf_fused_c4 lerf_metrics lerf_train lerf_transfer lerf_ubench; do objs="$objs $src/build/$o.o"; done
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o $src/build_variants/liblerf_hip_$name.so $objs $src/build_variants/li_$name.o
echo "built $src/build_variants/liblerf_hip_$name.so"
