#!/bin/bash
# lanes of a batch enqueue their later rounds just in time, looking every N us: headline leg (pageable) and resident frames
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs $1 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s, %.2f ms per step' % (d['value'], d['ms_per_step']))
"
}
for mem in "" "--host-memory device"; do
  echo "#### $mem"
  export LIBRECTIFY_FLOOD_PARTIAL_ROUNDS=3
  echo "== blind"; LIBRECTIFY_FLOOD_JIT_LANES=0 run "$mem"
  for n in 1 20 50 100; do echo "== just in time, pause $n us"; LIBRECTIFY_FLOOD_JIT_LANES=$n run "$mem"; done
  echo "== blind"; LIBRECTIFY_FLOOD_JIT_LANES=0 run "$mem"
done
