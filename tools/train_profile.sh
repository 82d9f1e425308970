#!/bin/bash
# Kernel-trace stats of the training-step benchmark (run through gpurun from the repo root).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 tools/train_bench.py --batch 32 --steps 3 --warmup 1 > $O/train.log 2>&1
tail -1 $O/train.log
find $O/train -name '*kernel_stats.csv' | xargs ls -la
