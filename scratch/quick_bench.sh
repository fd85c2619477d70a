#!/bin/bash
# scratch/quick_bench.sh TAG [bench args...]: one bench line without the CPU legs, key figures printed
TAG=$1; shift
mkdir -p gpurun_out/quick
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fresh-holders "$@" > gpurun_out/quick/$TAG.json 2>> gpurun_out/quick/err.log || { tail -5 gpurun_out/quick/err.log; exit 1; }
python - "$TAG" <<'PY'
import json, sys
d = json.load(open("gpurun_out/quick/%s.json" % sys.argv[1]))
print(sys.argv[1], d["value"], d["ms_per_step"], "launches", d.get("launches_per_step"), "host", d.get("host_issue_ms_per_step"))
for k in d.get("kernels", [])[:6]:
    print("   %-18s n=%3d %7.3f ms  %8.1f %s" % (k["kernel"], k["launches"], k["ms_per_step"], k["achieved"], k["unit"]),
          [(b["rows"][:9], b["launches"], b["ms_per_step"], b["achieved"]) for b in k.get("by_rows", [])[:3]])
PY
