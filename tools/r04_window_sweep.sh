#!/bin/bash
# staged window / hold-back / partial rounds with the logs on: single 4K frames
run() {
  timeout -k 10 200 python tools/run4k_seeds.py 1 2 3 4 1 2 3 4 2>&1 | python3 -c "
import sys,re
v=[];r=[]
for l in sys.stdin:
    m=re.search(r\"'flood_rounds': (\d+).*'log_give_ups': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: v.append(float(m.group(6))); r.append(int(m.group(1)))
print('   flood ms', v[4:], 'rounds', r[4:], 'mean %.3f' % (sum(v[4:])/max(len(v[4:]),1)))
"
}
echo "== default"; run
for w in "1,1" "1,2" "2,1" "2,2" "3,3"; do echo "== WINDOW=$w"; LIBRECTIFY_FLOOD_WINDOW=$w run; done
for h in 50 70 85; do echo "== HOLD=$h from the start"; LIBRECTIFY_FLOOD_HOLD=$h LIBRECTIFY_FLOOD_HOLD_START=1 run; done
for p in 0 1 2; do echo "== PARTIAL_ROUNDS=$p"; LIBRECTIFY_FLOOD_PARTIAL_ROUNDS=$p run; done
echo "== default"; run
