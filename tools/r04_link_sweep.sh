#!/bin/bash
# the timed leg of bench.py (64 pageable 4K frames a step) under settings that might matter to the host link
mkdir -p gpurun_out
run() { # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s value %8.1f Mpix/s  h2d %5.2f GB/s  cores: %s' % ('$name', r['value'], r['h2d']['GBps_per_rank'], r['host_cores_per_rank']['how']))"
}
run "default" X=1 --
run "one band a frame (34 MB)" LIBRECTIFY_BATCH_BAND_KB=34000 --
run "16 MB bands" LIBRECTIFY_BATCH_BAND_KB=16384 --
run "4 lanes" X=1 -- --streams 4
run "8 lanes" X=1 -- --streams 8
run "lanes sleep in their wait" LIBRECTIFY_LANES_SLEEP=1 --
run "no NUMA binding" X=1 -- --numa none
run "threads on the other node" X=1 -- --numa other
run "4 staging threads" X=1 -- --staging-threads 4
run "pinned frames" X=1 -- --host-memory pinned
