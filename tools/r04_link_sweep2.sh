#!/bin/bash
mkdir -p gpurun_out
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-52s value %8.1f Mpix/s  h2d %s' % ('$name', r['value'], (r.get('h2d') or {}).get('GBps_per_rank')))"
}
run "device-resident" X=1 -- --host-memory device
run "pinned" X=1 -- --host-memory pinned
run "pinned, copy stream at normal priority" LIBRECTIFY_COPY_STREAM_PLAIN=1 -- --host-memory pinned
run "pageable, copy stream at normal priority" LIBRECTIFY_COPY_STREAM_PLAIN=1 --
run "pinned, 16 hardware queues" GPU_MAX_HW_QUEUES=16 -- --host-memory pinned
run "pinned, 4 hardware queues" GPU_MAX_HW_QUEUES=4 -- --host-memory pinned
run "pinned, 7 lanes" X=1 -- --host-memory pinned --streams 7
run "device-resident, 7 lanes" X=1 -- --host-memory device --streams 7
