#!/bin/bash
# rounds just in time: off, lead 0, lead 1 (twice each, alternating)
for cfg in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1"; do
  set -- $cfg
  export LIBRECTIFY_FLOOD_JIT=$1 LIBRECTIFY_FLOOD_JIT_LEAD=$2
  echo "== LIBRECTIFY_FLOOD_JIT=$1 LEAD=$2"
  timeout -k 10 200 python tools/run4k_seeds.py 1 2 3 4 1 2 3 4 2>&1 | python3 -c "
import sys,re
v=[]
for l in sys.stdin:
    m=re.search(r\"total ([\d.]+) ms.*'log_give_ups': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: v.append(float(m.group(6)))
print('flood ms', v, 'mean %.3f' % (sum(v)/max(len(v),1)))
"
  timeout -k 10 200 python tools/run_doc4k.py 2>&1 | tail -1 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"'flood_rounds': (\d+).*\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: print('doc flood', m.group(5), 'ms rounds', m.group(1))
"
done
