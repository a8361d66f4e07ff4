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
for r in csv.DictReader(open(sys.argv[1])):
    if "lut_interp" in r["Name"]: print("%-110s calls %4s avg %9.1f us min %9.1f"%(r["Name"][:110],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3))
PY
