mkdir -p gpurun_out/fh
for cfg in "3 8" "2 8" "4 4" "6 2" "3 4" "2 4"; do set -- $cfg; for rep in 1 2; do JTVAE_FRESH_BATCHES=24 JTVAE_PREFETCH_WORKERS=$1 JTVAE_HOST_THREADS=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/fh/t_$1_$2_$rep.json 2>> gpurun_out/fh/err.log || exit 1; done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/fh/t_*.json")):
    d=json.load(open(f)); fh=d["fresh_holders"]
    print(f, d["value"], fh["value_fresh_holders_prefetched"], fh["value_from_flat_storage_prefetched"], fh["tensorize_and_prepare_host_ms_per_batch"])
PY
