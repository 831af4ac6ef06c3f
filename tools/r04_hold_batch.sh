#!/bin/bash
# the hold-back of the weakest seeds engaged from the first round in the lanes of a batch: less work, more rounds
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s, %.2f ms per step' % (d['value'], d['ms_per_step']))
"
}
echo "== as it is"; run
for h in 90 80 70 60; do echo "== HOLD=$h from the start"; LIBRECTIFY_FLOOD_HOLD=$h LIBRECTIFY_FLOOD_HOLD_START=1 run; done
echo "== as it is"; run
