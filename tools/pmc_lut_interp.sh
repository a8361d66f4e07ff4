#!/bin/bash
# SQ / LDS counters of the LDS-resident LUT pass (separate --pmc passes, no trace domains); usage: tools/pmc_lut_interp.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$i -o c -- python3 tools/bench_lut_interp.py "$@" > $out/pmc_$i.txt 2> $out/pmc_$i.err
done
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lut_interp" in r["Kernel_Name"]:
            nm = r["Kernel_Name"].split(">(")[0].replace("void lerf::(anonymous namespace)::", "").replace("void lerf::", "") + ">"
            acc[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s n=%3d mean %14.1f" % (c, len(v), sum(v) / len(v)))
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:          # KB units; FETCH_SIZE half-counts on gfx950 (MI355X_MICROARCH.md; calibrated 1.9 on this path)
        f, w = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) * 1024, sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024
        print("   HBM-side traffic per launch: fetch %.1f MB raw (x2 = %.1f MB), write %.1f MB" % (f / 1e6, 2 * f / 1e6, w / 1e6))
PY
