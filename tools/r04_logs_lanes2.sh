#!/bin/bash
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s, %.2f ms per step' % (d['value'], d['ms_per_step']))
"
}
for rep in 1 2 3; do
echo "== lanes without logs"; run
for m in "48 24" "64 32" "32 24"; do set -- $m; echo "== logs in lanes (LOG_MIN=$1 LOG_WALK=$2)"; LIBRECTIFY_FLOOD_LOGS_LANES=1 LIBRECTIFY_FLOOD_LOG_MIN=$1 LIBRECTIFY_FLOOD_LOG_WALK=$2 run; done
done
