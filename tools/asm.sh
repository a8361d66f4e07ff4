#!/bin/bash
# device assembly of the fused kernels + register summary: tools/asm.sh [out.s]
out=${1:-/tmp/fused.s}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -I/root/repo/include -S --cuda-device-only -o $out /root/repo/lerf-pytorch_amd/csrc/lerf_fused.hip $EXTRA 2>&1 | grep -v "warning: argument"
python3 - $out <<'PY'
import re,sys
t=open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', t):
    n=m.group(1)
    if 'fused' in n: print("%-75s sgpr %3s spill %3s vgpr %3s spill %3s" % (n[:75], m.group(2), m.group(3), m.group(4), m.group(5)))
PY
