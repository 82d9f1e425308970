#!/bin/bash
# Profile collection on the GPU box (run through gpurun from the repo root):  bash tools/gpu_profile.sh [TAG]
#   1. kernel-trace stats of the default bench command and of the single-stream variant,
#   2. separate counter passes (HBM fetch, HBM write) over the single-stream bench command,
#   3. matrix-pipe counters over the dominant conv shapes,
#   4. kernel-trace stats of one training configuration, in fp32 and with fp16x2 operands.
# Every GPU step has its own time limit and the script stops at the first failure.  The summaries are
# written under gpurun_out/prof/ and copied to gpurun_out/profiles_TAG/ under the names profiles/ uses.
set -e
export TMPDIR=/tmp
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
P=$R/gpurun_out/profiles_$TAG
rm -rf $O
mkdir -p $O $P
cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py > $O/bench.log 2>&1
grep -h '^{' $O/bench.log | tail -1 > $P/${TAG}_bench.json
cp $O/bench/bench_kernel_stats.csv $P/${TAG}_bench_kernel_stats.csv
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_g1 -o bench_g1 -- python3 bench.py --gops-in-flight 1 --cpu-baseline 0 --training 0 > $O/bench_g1.log 2>&1
grep -h '^{' $O/bench_g1.log | tail -1 > $P/${TAG}_bench_g1.json
cp $O/bench_g1/bench_g1_kernel_stats.csv $P/${TAG}_bench_g1_kernel_stats.csv
echo "bench_g1 done"
PMC_CMD="python3 bench.py --gops-in-flight 1 --cpu-baseline 0 --training 0 --alt-precision none --steps 1 --warmup 0 --gop 3"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- $PMC_CMD > $O/pmc_fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- $PMC_CMD > $O/pmc_write.log 2>&1
echo "write pass done"
python3 tools/pmc_summary.py $O/pmc_fetch/fetch_counter_collection.csv $O/pmc_write/write_counter_collection.csv \
  $P/${TAG}_pmc_traffic.json "$PMC_CMD" > $O/pmc_summary.log 2>&1
# MFMA pipe occupancy of the dominant conv shapes: busy cycles of the matrix pipe over all SIMDs
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o mfma -- python3 tools/conv_bench.py --shapes main,main,c128_64,k7_32_64 --iters 5 > $O/pmc_mfma.log 2>&1
python3 - "$O/pmc_mfma/mfma_counter_collection.csv" "$P/${TAG}_pmc_mfma.csv" <<'EOF'
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "conv_mfma" in k:
        agg[k[k.find("<"):k.find(">") + 1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    f.write("kernel,launches,counter,mean_per_launch\n")
    for k, d in agg.items():
        for c, v in d.items():
            f.write(f"\"conv_mfma_f32_kernel{k}\",{len(v)},{c},{sum(v) / len(v):.1f}\n")
EOF
echo "mfma pass done"
# the same counters for the 16-bit-operand (fp16 hi + lo) kernels of the dominant shapes: SQ_VALU_MFMA_BUSY_CYCLES counts
# 32 cycles per v_mfma_f32_32x32x16_f16
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma16 -o mfma16 -- python3 tools/conv_bench.py --precision 3 --shapes main,main,c128_64,half128_256,k7_32_64 --iters 5 > $O/pmc_mfma16.log 2>&1
python3 tools/pmc_mfma_summary.py "$O/pmc_mfma16/mfma16_counter_collection.csv" "$P/${TAG}_pmc_mfma_fp16x2.csv"
echo "mfma16 pass done"
timeout -k 10 200 python3 tools/conv_breakdown.py --top 60 > $P/${TAG}_conv_breakdown.txt 2>&1
timeout -k 10 200 python3 tools/conv_breakdown.py --precision fp16x2 --top 60 > $P/${TAG}_conv_breakdown_fp16x2.txt 2>&1
echo "breakdowns done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 tools/train_bench.py --batch 32 --frames 3 --size 256 --steps 3 --warmup 1 > $O/train.log 2>&1
grep -h '^{' $O/train.log | tail -1 > $P/${TAG}_train_b32.json
cp $O/train/train_kernel_stats.csv $P/${TAG}_train_b32_kernel_stats.csv
echo "train done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train16 -o train16 -- python3 tools/train_bench.py --batch 32 --frames 3 --size 256 --steps 3 --warmup 1 --precision fp16x2 > $O/train16.log 2>&1
grep -h '^{' $O/train16.log | tail -1 > $P/${TAG}_train_b32_fp16x2.json
cp $O/train16/train16_kernel_stats.csv $P/${TAG}_train_b32_fp16x2_kernel_stats.csv
echo "train fp16x2 done"
timeout -k 10 300 python3 tools/train_bench.py --batch 32 --frames 7 --size 256 --steps 2 --warmup 1 --recompute 1 > $O/train_t7.log 2>&1
grep -h '^{' $O/train_t7.log | tail -1 > $P/${TAG}_train_b32_t7_recompute.json
ls -la $P
# the raw traces are large (gpurun copies back at most 64 MiB): keep the logs, drop the rest
mkdir -p $P/logs && cp $O/*.log $P/logs/ 2>/dev/null || true
rm -rf $O
