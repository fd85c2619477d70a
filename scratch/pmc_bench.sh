#!/bin/bash
# HBM traffic counters of the bench's kernels: one rocprofv3 pass per counter group
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$R/gpurun_out/pmc_bench_$c
  rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  ls $OUT/*/ | head
done
