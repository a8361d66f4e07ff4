#!/bin/bash
# per-kernel times of tools/bench_lut_interp.py under rocprofv3 (kernel trace only); usage: tools/prof_lut_interp.sh <tag> [args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 tools/bench_lut_interp.py "$@" > $out/bench.txt 2> $out/bench.err
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
import re
# roofline of the call the tool makes (a float32 HWC frame of 3 channels in, [3 oC, h, w] out): algorithmic bytes = frame in + values
# out (+ the planes read when accumulating) over the kernel's average duration, against 8 TB/s
hw = 1080 * 1920
for a in sys.argv[2:]:
    pass
for r in csv.DictReader(open(sys.argv[1])):
    if "lut_interp" not in r["Name"]: continue
    us = float(r["AverageNs"]) / 1e3
    m = re.search(r"lut_interp(?:_lds)?_kernel<(\d), (\w+(?: \w+)?), (\w+), (true|false)", r["Name"])
    roof = ""
    if m:
        oC, tout, acc = int(m.group(1)), m.group(3), m.group(4) == "true"
        eb = {"double": 8, "float": 4, "short": 2}.get(tout, 8)
        b = hw * 3 * 4 + hw * 3 * oC * eb * (2 if acc else 1)
        roof = "  %6.1f MB -> %5.2f TB/s = %4.1f %% of 8 TB/s" % (b / 1e6, b / us / 1e6, 100 * b / us / 1e6 / 8.0)
    print("%-104s calls %4s avg %7.1f us min %7.1f%s" % (r["Name"][:104], r["Calls"], us, float(r["MinNs"]) / 1e3, roof))
PY
