#!/bin/bash
# Compiles every HIP source of the library to gfx950 assembly (device only) and lists the kernels that use scratch memory or
# spill registers -- the tile-fused kernels are sized to need none (a spill in a stage-3 task shows up as doubled WRITE_SIZE
# long before it shows in the timing).  usage: tools/check_spills.sh [extra hipcc flags]     (CPU only, ~1 min)
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/lerf-pytorch_amd/csrc
tmp=$(mktemp -d)
for f in $src/*.hip; do
  b=$(basename $f .hip)
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -fno-slp-vectorize -w --cuda-device-only -S $* -o $tmp/$b.s $f 2>/dev/null ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
python3 - $tmp <<'PY'
import glob, re, sys
bad = 0
for f in sorted(glob.glob(sys.argv[1] + "/*.s")):
    t = open(f).read()
    for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", t, re.S):
        name, scratch, vg, sp = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
        if scratch or sp:
            bad += 1
            print("%-22s scratch %4d B  vgprs %3d  spilled %3d  %s" % (f.split("/")[-1], scratch, vg, sp, name[:90]))
print("%d kernel(s) with scratch or spills" % bad)
PY
rm -rf $tmp
