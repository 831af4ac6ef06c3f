#!/bin/bash
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs $1 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s, %.2f ms per step' % (d['value'], d['ms_per_step']))
"
}
for rep in 1 2; do
for s in 4 5 6 7; do echo "== $s lanes"; run "--streams $s"; done
for m in "24 16" "32 16" "40 24" "32 32"; do set -- $m; echo "== lanes' logs MIN=$1 WALK=$2"; LIBRECTIFY_FLOOD_LOG_MIN_LANES=$1 LIBRECTIFY_FLOOD_LOG_WALK_LANES=$2 run; done
done
