#!/bin/bash
# Round-4 profiling pass on the GPU box: bash tools/profile_r04.sh   -> gpurun_out/prof_r04_*/ (copied to profiles/r04/)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
set -x
bash tools/profile.sh r04_lean --no-configs || exit 1
bash tools/profile.sh r04_nsf --workload nsf64 || exit 1
bash tools/profile.sh r04_rnvp256 --workload realnvp256 || exit 1
bash tools/profile.sh r04_glow32 --workload glow32 || exit 1
# the inverse (Flow.sample) program and the training step: kernel traces
mkdir -p gpurun_out/prof_r04_sample gpurun_out/prof_r04_train
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_sample/trace -o trace -- python bench.py --steps 2 --warmup 1 --stats-steps 0 --priming 0 --no-cpu-baseline --no-train --no-configs > gpurun_out/prof_r04_sample/bench.json 2> gpurun_out/prof_r04_sample/err.txt
python tools/summarize_profile.py gpurun_out/prof_r04_sample > /dev/null; rm -rf gpurun_out/prof_r04_sample/trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_train/trace -o trace -- python tools/train_probe.py > gpurun_out/prof_r04_train/probe.txt 2> gpurun_out/prof_r04_train/err.txt
python tools/summarize_profile.py gpurun_out/prof_r04_train > /dev/null; rm -rf gpurun_out/prof_r04_train/trace
# config 5's model in TRAINING: the captured step of Flow.fit (ConvNet conditioner on csrc/tfk_convtrain.hip)
mkdir -p gpurun_out/prof_r04_train_glow
TORCHFLOWS_AMD_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_train_glow/trace -o trace -- python tools/image_graph_probe.py 60 glow > gpurun_out/prof_r04_train_glow/probe.txt 2> gpurun_out/prof_r04_train_glow/err.txt
python tools/summarize_profile.py gpurun_out/prof_r04_train_glow > /dev/null; rm -rf gpurun_out/prof_r04_train_glow/trace
python tools/convtrain_bench.py 1024 > gpurun_out/prof_r04_train_glow/convtrain_bench.txt 2>/dev/null
bash tools/convtrain_pmc.sh r04 8192 > gpurun_out/prof_r04_train_glow/pmc_8192.txt 2>&1
bash tools/convtrain_pmc.sh r04b 1024 > gpurun_out/prof_r04_train_glow/pmc_1024.txt 2>&1
# one coupling launch of config 5 (the first checkerboard layer) under the SQ / TCC counters
bash tools/glow_pmc.sh r04_step0 0 131072 > gpurun_out/prof_r04_glow32/step0_pmc.txt 2>&1
bash tools/glow_pmc.sh r04_step3 3 131072 > gpurun_out/prof_r04_glow32/step3_pmc.txt 2>&1
echo PROFILE_DONE
