#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):
#   1. kernel-trace stats of the default bench command and of the single-stream variant,
#   2. separate counter passes (HBM fetch, HBM write) over the single-stream bench command,
#   3. SQ cycle counters over the dominant conv shapes.
# Every GPU step has its own time limit and the script stops at the first failure.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py > $O/bench.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_g1 -o bench_g1 -- python3 bench.py --gops-in-flight 1 --cpu-baseline 0 > $O/bench_g1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 bench.py --gops-in-flight 1 --cpu-baseline 0 --alt-precision none --steps 1 --warmup 0 --gop 3 > $O/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 bench.py --gops-in-flight 1 --cpu-baseline 0 --alt-precision none --steps 1 --warmup 0 --gop 3 > $O/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq -o sq -- python3 tools/conv_bench.py --shapes main,c128_64 --iters 5 > $O/pmc_sq.log 2>&1
find $O -name '*.csv' | xargs ls -la
# MFMA pipe occupancy of the dominant conv shapes: busy cycles of the matrix pipe over all SIMDs
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o mfma -- python3 tools/conv_bench.py --shapes main,main,c128_64,k7_32_64 --iters 5 > $O/pmc_mfma.log 2>&1
find $O/pmc_mfma -name '*.csv' | xargs ls -la
