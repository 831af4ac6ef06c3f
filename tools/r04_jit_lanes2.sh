#!/bin/bash
# lanes just in time (pause 20 us) x logs in the lanes x lanes in flight: headline leg
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs $1 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s, %.2f ms per step' % (d['value'], d['ms_per_step']))
"
}
export LIBRECTIFY_FLOOD_PARTIAL_ROUNDS=3 LIBRECTIFY_FLOOD_JIT_LANES=20
echo "== jit lanes, logs off in lanes"; run
echo "== jit lanes, logs on in lanes (16/12)"; LIBRECTIFY_FLOOD_LOGS_LANES=1 run
echo "== jit lanes, logs on in lanes (48/24)"; LIBRECTIFY_FLOOD_LOGS_LANES=1 LIBRECTIFY_FLOOD_LOG_MIN=48 LIBRECTIFY_FLOOD_LOG_WALK=24 run
for s in 5 7 8; do echo "== jit lanes, $s streams"; run "--streams $s"; done
echo "== jit lanes, 8 streams, logs in lanes"; LIBRECTIFY_FLOOD_LOGS_LANES=1 run "--streams 8"
