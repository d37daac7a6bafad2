#!/bin/bash
# Profiling recipe run on the GPU box (see profiles/README.md).
# Usage: bash tools/profile.sh <tag> [bench.py args]   -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-r01}; shift || true
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# 1) kernel trace + stats (per-kernel durations)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 10 --warmup 2 --stats-steps 0 --no-cpu-baseline --no-sample --no-train "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
# 2) HBM read bytes (FETCH_SIZE needs 3 TCC slots), 3) write bytes, 4) VALU / LDS activity: separate passes
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python bench.py --steps 3 --warmup 1 --stats-steps 0 --no-cpu-baseline --no-sample --no-train "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python bench.py --steps 3 --warmup 1 --stats-steps 0 --no-cpu-baseline --no-sample --no-train "$@" > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python bench.py --steps 3 --warmup 1 --stats-steps 0 --no-cpu-baseline --no-sample --no-train "$@" > $OUT/bench_sq.json 2> $OUT/sq.err
# 5) matrix-core counters (the MFMA instruction count bench.py models analytically, matrix-pipe busy cycles, cycles
#    in which a vector instruction issued while the matrix pipe was busy), 6) the mix of the vector instructions
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU2 --kernel-trace --output-format csv -d $OUT/pmc_mfma -o mfma -- python bench.py --steps 3 --warmup 1 --stats-steps 0 --no-cpu-baseline --no-sample --no-train "$@" > $OUT/bench_mfma.json 2> $OUT/mfma.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS_LOAD --kernel-trace --output-format csv -d $OUT/pmc_mix -o mix -- python bench.py --steps 3 --warmup 1 --stats-steps 0 --no-cpu-baseline --no-sample --no-train "$@" > $OUT/bench_mix.json 2> $OUT/mix.err
python tools/summarize_profile.py $OUT
# raw traces are large (gpurun_out/ merges back at most 64 MiB): keep the summaries only
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_mfma $OUT/pmc_mix
