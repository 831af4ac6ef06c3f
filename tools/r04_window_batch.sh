#!/bin/bash
# staged window in the lanes of a batch, now that late rounds are cheap (just in time, fewer launches)
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs $1 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s, %.2f ms per step' % (d['value'], d['ms_per_step']))
"
}
echo "== no window"; run
for w in "1,1" "1,2" "2,2" "3,2" "3,3"; do echo "== WINDOW=$w"; LIBRECTIFY_FLOOD_WINDOW=$w run; done
echo "== no window"; run
