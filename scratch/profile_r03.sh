#!/bin/bash
# rocprofv3 passes over one bench workload: kernel trace + stats, then one --pmc pass per counter group.
#   scratch/profile_r03.sh MODE BATCH [stats|stats1|pmc|all]
# Outputs under gpurun_out/r03_<mode>_<batch>/; the summaries to keep are copied to profiles/ afterwards.
MODE=${1:-train}; BATCH=${2:-1024}; WHAT=${3:-all}
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_${MODE}_${BATCH}
mkdir -p $OUT
if [ $WHAT = stats ] || [ $WHAT = all ]; then
  D=$OUT/stats; rm -rf $D; mkdir -p $D
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --mode $MODE --batch $BATCH --steps 10 --warmup 3 --no-cpu-baseline --no-fresh-holders > $D/bench.json 2> $D/bench.err || { tail -5 $D/bench.err; exit 1; }
  cp $D/*/*_kernel_stats.csv $OUT/kernel_stats.csv || exit 1
fi
if [ $WHAT = stats1 ]; then  # the same command on one stream: per-kernel durations free of co-running kernels
  D=$OUT/stats1; rm -rf $D; mkdir -p $D
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --mode $MODE --batch $BATCH --steps 10 --warmup 3 --no-overlap --no-cpu-baseline --no-fresh-holders > $D/bench.json 2> $D/bench.err || { tail -5 $D/bench.err; exit 1; }
  cp $D/*/*_kernel_stats.csv $OUT/kernel_stats_one_stream.csv || exit 1
fi
if [ $WHAT = pmc ] || [ $WHAT = all ]; then
  for c in FETCH_SIZE WRITE_SIZE SQ; do
    D=$OUT/pmc_$c; rm -rf $D; mkdir -p $D
    CNT=$c
    [ $c = SQ ] && CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
    timeout -s KILL 240 rocprofv3 --pmc $CNT --output-format csv -d $D -- python3 $R/bench.py --mode $MODE --batch $BATCH --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fresh-holders > $D/bench.json 2> $D/bench.err || { tail -5 $D/bench.err; exit 1; }
  done
  cd $R
  python3 scratch/pmc_summarise.py traffic $OUT $MODE $BATCH 7 gpurun_out/r03_pmc_traffic.json || exit 1
  python3 scratch/pmc_summarise.py sq $OUT $MODE $BATCH 7 gpurun_out/r03_sq_counters.json || exit 1
  # the counter CSVs are large: keep the summaries only
  rm -rf $OUT/pmc_FETCH_SIZE/*/ $OUT/pmc_WRITE_SIZE/*/ $OUT/pmc_SQ/*/
fi
