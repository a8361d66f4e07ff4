#!/bin/bash
# Full profile of the headline bench (run on the GPU box through gpurun): kernel trace + stats, then one rocprofv3 --pmc
# pass per counter group (never combined with the trace domains).  Output: gpurun_out/prof_<tag>/{stats,pmc_*}.
#   usage: tools/profile.sh <tag> [extra bench.py args]      then here: python tools/summarize_profile.py <tag> <label>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-input --sustained 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- $B > $out/bench.json 2> $out/bench.err
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$i -o c -- $B > $out/pmc_$i.json 2> $out/pmc_$i.err
done
find $out -name "*.csv" | head -20
tail -1 $out/bench.json | cut -c1-300
