#!/bin/bash
# kernel trace of one rank's batched block launch pair: experiments/probes/probe_blocks.sh <rank> ...
cd /tmp && export TMPDIR=/tmp
for r in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pb$r -o k -- python3 $GRAFT_REPO_ROOT/experiments/probes/probe_blocks.py $r 2>&1 | grep rank
  python3 - $GRAFT_REPO_ROOT/gpurun_out/pb$r <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    if "s1_kernel" in row["Name"] or "sr_fused" in row["Name"]:
        print("   %-70s calls %s avg %.1f us" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
done
