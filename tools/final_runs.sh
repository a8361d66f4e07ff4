#!/bin/bash
# every measured artefact of a round in one gpurun call: tools/final_runs.sh <tag>  (outputs under gpurun_out/<tag>/ and gpurun_out/prof_*)
tag=${1:-r03}
mkdir -p gpurun_out/$tag
python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
{
for a in "--config 1" "--config 3" "--config 4" "--config 5 --frames 4" "--support 4" "--scale 6" "--scale 8" "--channels 1" "--channels 4" "--input natural"; do
  echo "== bench.py $a"
  python bench.py $a --steps 20 --warmup 3 --sustained 2 2>/dev/null | tail -1
done
} > gpurun_out/$tag/other_configs.txt
python tools/bench_8k.py > gpurun_out/$tag/8k_blocks_strips.txt 2>&1
python tools/bench_ragged.py > gpurun_out/$tag/ragged_set5.txt 2>&1
python tools/stamps.py noise natural > gpurun_out/$tag/phase_breakdown.txt 2>&1
./tools/ubench/lds_gather > gpurun_out/$tag/lds_gather.txt 2>&1
python tools/eval_set5.py > gpurun_out/$tag/set5_table.txt 2>&1
bash tools/prof_all.sh > gpurun_out/$tag/prof_all.log 2>&1
for c in "c2 " "c2s4 --support 4" "c3 --config 3" "c4 --config 4" "c5 --config 5 --frames 4"; do set -- $c; bash tools/prof_quick.sh q$1 ${@:2} > gpurun_out/$tag/kernels_$1.txt 2>&1; done
ls gpurun_out/$tag
