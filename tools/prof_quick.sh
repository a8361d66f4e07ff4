#!/bin/bash
# quick per-kernel timing of the headline bench under rocprofv3 (kernel trace only); usage: tools/prof_quick.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-input --sustained 0 "$@" > $out/bench.json 2> $out/bench.err
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"])>0.5: print("%-90s calls %4s avg %9.1f us  %5.1f%%"%(r["Name"][:90],r["Calls"],float(r["AverageNs"])/1e3,float(r["Percentage"])))
PY
tail -1 $out/bench.json | cut -c1-200
