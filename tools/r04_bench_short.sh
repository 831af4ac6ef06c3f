#!/bin/bash
# the default bench line (short) and its headline keys
tag=${1:-x}; shift
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --steps 6 --warmup 2 "$@" > gpurun_out/r04_bench_$tag.json 2> gpurun_out/r04_bench_$tag.err; echo "bench rc $?"
python3 - gpurun_out/r04_bench_$tag.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms/step", r["ms_per_step"], "h2d", r["h2d"]["GBps_per_rank"])
print("other", {k: v for k, v in r["other_rates_Mpix_per_s"].items() if k != "note"})
for k in ("single_frame", "flood", "natural_frame", "worst_case", "stage_ms_per_frame"):
    print(k, json.dumps(r.get(k))[:600])
if r.get("cpu_baseline"): print("cpu", json.dumps(r["cpu_baseline"])[:700])
PY
