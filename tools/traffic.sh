#!/bin/bash
# HBM-side traffic of the headline bench only (two rocprofv3 --pmc passes): tools/traffic.sh <tag> ; pass `--lib PATH` among the bench arguments to profile another build
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/traffic_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-input --sustained 0 $*"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -o c -- $B > /dev/null 2> $out/p1.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -o c -- $B > /dev/null 2> $out/p2.err
python3 - $out <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/p*/**/*counter_collection.csv", recursive=True):
    rows=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k="s1" if "s1_kernel" in r["Kernel_Name"] else ("s23" if "sr_fused_kernel" in r["Kernel_Name"] else ("warp" if "warp_packed" in r["Kernel_Name"] else None))
        if k: rows[(k,r["Dispatch_Id"])][r["Counter_Name"]]+=float(r["Counter_Value"])
    for (k,_),cs in rows.items():
        for c,v in cs.items(): acc[k][c].append(v)
for k in sorted(acc):
    d={c:sum(v)/len(v) for c,v in acc[k].items()}
    print(k, "FETCH %.1f MB  WRITE %.1f MB  L2 hit %.3f" % (d.get("FETCH_SIZE",0)*1024/1e6, d.get("WRITE_SIZE",0)*1024/1e6, d.get("TCC_HIT_sum",0)/max(1,d.get("TCC_HIT_sum",0)+d.get("TCC_MISS_sum",0))), {c:round(v) for c,v in d.items() if c!="GRBM_GUI_ACTIVE"})
PY
