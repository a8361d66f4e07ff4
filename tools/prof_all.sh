set -x
bash tools/profile.sh c2 > gpurun_out/prof_c2.log 2>&1
bash tools/profile.sh c2s4 --support 4 > gpurun_out/prof_c2s4.log 2>&1
bash tools/profile.sh c3 --config 3 > gpurun_out/prof_c3.log 2>&1
bash tools/profile.sh c4 --config 4 > gpurun_out/prof_c4.log 2>&1
bash tools/profile.sh c4f --config 4 --warp-fused > gpurun_out/prof_c4f.log 2>&1
bash tools/profile.sh c5 --config 5 --frames 4 > gpurun_out/prof_c5.log 2>&1
bash tools/profile.sh c1 --channels 1 > gpurun_out/prof_c1.log 2>&1
ls gpurun_out
