#!/bin/bash
# usage (on the GPU box): tools/sweep_streams.sh "<hw queues...>" "<lanes...>" [extra bench args]
# throughput of bench.py's timed region over HIP hardware queues (GPU_MAX_HW_QUEUES) x frames in flight
mkdir -p gpurun_out/sweep
for q in $1; do for s in $2; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --streams $s $3 > gpurun_out/sweep/b_q${q}_s${s}.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("gpurun_out/sweep/b_q${q}_s${s}.json"))
print("queues $q lanes $s: %.0f Mpix/s" % d["value"], d["stage_ms_per_frame"])
PY
done; done
