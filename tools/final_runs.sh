#!/bin/bash
# every measured artefact of a round in one gpurun call: tools/final_runs.sh <tag>  (outputs under gpurun_out/<tag>/ and gpurun_out/prof_*)
tag=${1:-r06}
mkdir -p gpurun_out/$tag
timeout 900 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
{
for a in "--config 1" "--config 3" "--config 4" "--config 4 --warp-fused" "--config 5 --frames 4" "--support 4" "--scale 6" "--scale 8" "--channels 1" "--channels 4" "--input natural"; do
  echo "== bench.py $a"
  timeout 300 python bench.py $a --steps 20 --warmup 3 --sustained 2 2>/dev/null | tail -1
done
} > gpurun_out/$tag/other_configs.txt
timeout 400 python tools/bench_8k.py > gpurun_out/$tag/8k_blocks_strips.txt 2>&1
timeout 400 python tools/bench_ragged.py > gpurun_out/$tag/ragged_set5.txt 2>&1
timeout 400 python tools/stamps.py noise natural > gpurun_out/$tag/phase_breakdown.txt 2>&1
timeout 400 python bench.py --path callsite > gpurun_out/$tag/callsite.json 2>/dev/null
timeout 400 python bench.py --path classes-torch > gpurun_out/$tag/classes_torch.json 2>/dev/null
timeout 400 python experiments/probes/probe_warp.py > gpurun_out/$tag/warp_split.txt 2>&1
timeout 400 python tools/eval_set5.py > gpurun_out/$tag/set5_table.txt 2>&1
timeout 1500 bash tools/prof_all.sh > gpurun_out/$tag/prof_all.log 2>&1
# the single LUT pass behind the reference's own signature (lerf_lut_interp_ex): kernel times per (oC, pattern, rotation), LDS kernel and direct kernel
timeout 300 bash tools/prof_lut_interp.sh ${tag}_lut_interp --planes > gpurun_out/$tag/lut_interp_kernels.txt 2>&1
cp gpurun_out/prof_${tag}_lut_interp/bench.txt gpurun_out/$tag/lut_interp_calls.txt
timeout 700 bash tools/pmc_lut_interp.sh ${tag} --quick --kernels lds --iters 3 --no-acc --planes > gpurun_out/$tag/lut_interp_pmc.txt 2>&1
timeout 300 bash tools/prof_callsite.sh ${tag}_callsite > gpurun_out/$tag/callsite_profile.txt 2>&1
ls gpurun_out/$tag
# direct kernels behind the unchanged call sites (FourSimplexInterpFaster x 24, resize): kernel trace of bench.py --path callsite
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag/callsite_trace -o k -- python3 $GRAFT_REPO_ROOT/bench.py --path callsite --steps 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
