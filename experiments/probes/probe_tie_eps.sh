#!/bin/bash
# A/B of the tie guard: adversarial parity + throughput per build.  The product detects at 1.5e-4; the strict build
# (tools/build_variant_all.sh strict "-DLERF_TIE_EPS=1e-3f") at 1e-3; both resolve in two levels (s3::resolve_u8).
V=lerf-pytorch_amd/csrc/build_variants
LIBS=${LIBS:-"$V/liblerf_hip_strict.so"}
for lib in "" $LIBS; do
  echo "=== ${lib:-product}"
  timeout 200 python experiments/probes/probe_tie_eps.py $lib 2>&1 | grep -v amdgpu
done
for rep in 1 2 3; do
  for lib in "" $LIBS; do
    for cfg in "--input noise" "--input natural" "--support 4" "--config 3" "--config 4"; do
      if [ -z "$lib" ]; then a=""; else a="--lib $lib"; fi
      echo -n "${lib:-product} $cfg: "
      timeout 200 python bench.py $a --steps 60 --warmup 5 --no-cpu-baseline --no-other-input --sustained 0 --no-psnr $cfg 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
    done
  done
done
