#!/bin/bash
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s value %8.1f Mpix/s' % ('$name', r['value']))"; }
run "pinned, SDMA engines (default)" X=1 -- --host-memory pinned
run "pinned, HSA_ENABLE_SDMA=0 (copies by shader kernels)" HSA_ENABLE_SDMA=0 -- --host-memory pinned
run "pageable, HSA_ENABLE_SDMA=0" HSA_ENABLE_SDMA=0 --
run "device-resident, HSA_ENABLE_SDMA=0" HSA_ENABLE_SDMA=0 -- --host-memory device
