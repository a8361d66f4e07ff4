#!/bin/bash
# A/B of the tie-guard epsilon: adversarial parity + headline / natural-input throughput per build
V=lerf-pytorch_amd/csrc/build_variants
for lib in "" $V/liblerf_hip_eps1e3.so $V/liblerf_hip_eps3e3.so; do
  echo "=== ${lib:-product (1.5e-4)}"
  timeout 200 python experiments/probes/probe_tie_eps.py $lib 2>&1 | grep -v amdgpu
done
for rep in 1 2; do
  for lib in "" $V/liblerf_hip_eps1e3.so $V/liblerf_hip_eps3e3.so; do
    for inp in noise natural; do
      if [ -z "$lib" ]; then a=""; else a="--lib $lib"; fi
      echo -n "${lib:-product} $inp: "
      timeout 200 python bench.py $a --steps 60 --warmup 5 --no-cpu-baseline --no-other-input --sustained 0 --no-psnr --input $inp 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity_vs_cpu_port'))"
    done
  done
done
