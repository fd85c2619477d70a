# the three whole-step bench lines again (after their counter summary has been merged into profiles/)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_bench
timeout -k 10 300 python bench.py --mode train_full --batch 512 --no-cpu-baseline > gpurun_out/r03_bench/train_full_512.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 600 python bench.py --mode train_full --batch 4096 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench/train_full_4096.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 300 python bench.py --mode train_prop --batch 512 --no-cpu-baseline > gpurun_out/r03_bench/train_prop_512.json 2>> gpurun_out/r03_bench/err.log || exit 1
timeout -k 10 300 python bench.py --mode train_assm --no-cpu-baseline > gpurun_out/r03_bench/train_assm_1024.json 2>> gpurun_out/r03_bench/err.log || exit 1
echo collected
