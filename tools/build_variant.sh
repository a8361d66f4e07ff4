#!/bin/bash
# Builds a VARIANT of the library for A/B runs on one GPU box (tools/ab.sh): lerf_fused.hip is recompiled with extra flags
# and, optionally, its device assembly re-encoded by tools/e64.py; every other object is taken from csrc/build.
#   tools/build_variant.sh NAME ["extra hipcc flags"] ["e64.py options" | none]
# -> lerf-pytorch_amd/csrc/build_variants/liblerf_hip_NAME.so   (git-ignored; select it with `bench.py --lib PATH` / _lib.use_library)
set -e
name=$1; extra=$2; e64=${3:-none}
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/lerf-pytorch_amd/csrc
LL=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d /tmp/variant_$name.XXXX)
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -Wno-unused-function -fno-slp-vectorize -w $extra"
make -s -C $src >/dev/null
/opt/rocm/bin/hipcc $flags -S --cuda-device-only -o $tmp/dev.s $src/lerf_fused.hip
if [ "$e64" != "none" ]; then python3 $root/tools/e64.py $tmp/dev.s $tmp/dev2.s $e64 --stats; else cp $tmp/dev.s $tmp/dev2.s; fi
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $tmp/dev2.s -o $tmp/dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $tmp/dev.hsaco $tmp/dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input=$tmp/dev.hsaco -output=$tmp/dev.hipfb
/opt/rocm/bin/hipcc $flags --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $tmp/dev.hipfb -c -o $tmp/lerf_fused.o $src/lerf_fused.hip
objs=""
for o in lerf_api lerf_kernels lerf_fused_g3 lerf_fused_h32 lerf_fused_h16 lerf_fused_c1 lerf_fused_c4 lerf_metrics lerf_train lerf_transfer lerf_ubench; do objs="$objs $src/build/$o.o"; done
mkdir -p $src/build_variants
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o $src/build_variants/liblerf_hip_$name.so $objs $tmp/lerf_fused.o
cp $tmp/dev2.s /tmp/variant_$name.s
rm -rf $tmp
echo "built lerf-pytorch_amd/csrc/build_variants/liblerf_hip_$name.so (assembly kept in /tmp/variant_$name.s)"
