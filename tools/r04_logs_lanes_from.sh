#!/bin/bash
# logs in the lanes of a batch, used only from round N on (walks leave them from the first round): headline leg
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s, %.2f ms per step' % (d['value'], d['ms_per_step']))
"
}
echo "== lanes without logs"; run
for f in 1 2 3; do for m in "16 12" "48 24"; do set -- $m; echo "== logs in lanes from round $f (LOG_MIN=$1 LOG_WALK=$2)"; LIBRECTIFY_FLOOD_LOGS_LANES=1 LIBRECTIFY_FLOOD_LOGS_LANES_FROM=$f LIBRECTIFY_FLOOD_LOG_MIN=$1 LIBRECTIFY_FLOOD_LOG_WALK=$2 run; done; done
echo "== lanes without logs"; run
