#!/bin/bash
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s value %8.1f Mpix/s' % ('$name', r['value']))"; }
for e in 1 2 3 6 12; do run "pinned, pool = lanes + $e" LIBRECTIFY_RING_EXTRA=$e -- --host-memory pinned; done
run "pageable, pool = lanes + 2" LIBRECTIFY_RING_EXTRA=2 --
