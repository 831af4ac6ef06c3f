#!/bin/bash
# headline leg per lane count (device-resident frames unless MEM is set): usage tools/exp_lanes.sh <out> <lanes...>
out=$1; shift; mkdir -p gpurun_out; : > gpurun_out/$out
for l in "$@"; do
  for rep in 1 2; do
    v=$(python3 bench.py --no-cpu-baseline --no-extra-legs --steps 10 --warmup 3 --host-memory ${MEM:-device} --streams $l 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['stage_ms_per_frame'])")
    echo "lanes $l | $v" >> gpurun_out/$out
  done
done
cat gpurun_out/$out
