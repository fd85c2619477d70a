#!/bin/bash
# Every measurement kept under profiles/ for this round, in two GPU calls:
#   collect_r03.sh profile: rocprofv3 kernel stats + PMC passes per workload;  collect_r03.sh bench: the plain
#   bench lines (run after the counter summary of the first call has been copied to profiles/).
set -o pipefail
R=$GRAFT_REPO_ROOT
if [ "$1" = profile ]; then
for w in "train 1024" "train 4096" "encode 1024" "encode 4096" "train_assm 1024" "train_full 512" "train_full 4096" "train_prop 512"; do
  set -- $w
  bash $R/scratch/profile_r03.sh $1 $2 all || exit 1
done
bash $R/scratch/profile_r03.sh train 1024 stats1 || exit 1
echo profiled
exit 0
fi
# second call (after gpurun_out/r03_pmc_traffic.json has been copied to profiles/): the plain bench lines
cd $R
mkdir -p gpurun_out/r03_bench
timeout -k 10 300 python bench.py > gpurun_out/r03_bench/train_1024.json 2> gpurun_out/r03_bench/train_1024.err || exit 1
timeout -k 10 300 python bench.py --no-overlap --no-cpu-baseline --no-fresh-holders > gpurun_out/r03_bench/train_1024_one_stream.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 300 python bench.py --batch 256 --no-cpu-baseline --no-fresh-holders > gpurun_out/r03_bench/train_256.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 300 python bench.py --batch 4096 --no-cpu-baseline --no-fresh-holders > gpurun_out/r03_bench/train_4096.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 300 python bench.py --mode encode > gpurun_out/r03_bench/encode_1024.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 300 python bench.py --mode encode --batch 4096 > gpurun_out/r03_bench/encode_4096.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 300 python bench.py --mode train_assm --no-cpu-baseline > gpurun_out/r03_bench/train_assm_1024.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 300 python bench.py --mode train_full --batch 512 --no-cpu-baseline > gpurun_out/r03_bench/train_full_512.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 600 python bench.py --mode train_full --batch 4096 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench/train_full_4096.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 300 python bench.py --mode train_prop --batch 512 --no-cpu-baseline > gpurun_out/r03_bench/train_prop_512.json 2>> gpurun_out/r03_bench/err.log || exit 1
echo collected
