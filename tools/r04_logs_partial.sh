#!/bin/bash
# logs on: partial-commit launches only in the first N rounds
export LIBRECTIFY_FLOOD_LOGS=1 LIBRECTIFY_FLOOD_LOG_MIN=16
for n in 1000 5 4 3 2; do
  echo "== PARTIAL_ROUNDS=$n"
  LIBRECTIFY_FLOOD_PARTIAL_ROUNDS=$n timeout -k 10 200 python tools/run4k_seeds.py 2>&1 | python3 -c "
import sys,re
v=[];r=[]
for l in sys.stdin:
    m=re.search(r\"'flood_rounds': (\d+).*'log_rewalks': (\d+), 'log_give_ups': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: v.append(float(m.group(7))); r.append(int(m.group(1)))
print('flood ms', v, 'rounds', r, 'mean %.3f' % (sum(v)/max(len(v),1)))
"
done
