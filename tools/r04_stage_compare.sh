#!/bin/bash
# per-stage device times of frames inside the six-lane pipeline, by where the frames come from
for kind in device pinned pageable; do
  python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline --host-memory $kind 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-9s value %8.1f Mpix/s  stages %s' % ('$kind', r['value'], json.dumps(r['stage_ms_per_frame'])))"
done
