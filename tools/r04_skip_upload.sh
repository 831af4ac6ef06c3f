#!/bin/bash
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s value %8.1f Mpix/s' % ('$name', r['value']))"; }
run "device-resident" X=1 -- --host-memory device
run "pinned" X=1 -- --host-memory pinned
run "pinned, transfers skipped (events only, stale frames)" LIBRECTIFY_EXPERIMENT_SKIP_UPLOAD=1 -- --host-memory pinned
run "pageable, staging + transfers skipped" LIBRECTIFY_EXPERIMENT_SKIP_UPLOAD=1 --
