#!/bin/bash
# per-kernel device times of bench.py --path callsite under rocprofv3 (kernel trace only); usage: tools/prof_callsite.sh <tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py --path callsite --steps 20 "$@" > $out/callsite.json 2> $out/callsite.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 bench.py --path callsite --steps 20 "$@" > $out/bench.json 2> $out/bench.err
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-100s calls %5s avg %8.1f us total %8.2f ms %5.1f%%"%(r["Name"][:100],r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6,float(r["Percentage"])))
print("total device ms", tot/1e6)
PY
python3 -c "
import json,sys
d=json.loads(open('$out/callsite.json').read().strip().splitlines()[-1])
for k,v in d['legs'].items(): print(k, v)
"
