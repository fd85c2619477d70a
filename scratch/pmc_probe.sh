#!/bin/bash
# SQ counters of the split-product probe: scratch/pmc_probe.sh BINARY
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=$1
OUT=$R/gpurun_out/pmc_probe; rm -rf $OUT; mkdir -p $OUT
WARM=3 timeout -s KILL 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/a -- $R/scratch/bin/$B 85181 > $OUT/a.log 2>&1 || { tail -3 $OUT/a.log; exit 1; }
WARM=3 timeout -s KILL 60 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- $R/scratch/bin/$B 85181 > $OUT/b.log 2>&1 || { tail -3 $OUT/b.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "nt_" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-28s %14.0f per launch" % (k, tot[k] / max(n[k], 1)))
PY
