#!/bin/bash
# how many rounds a frame enqueues blindly before it goes round by round: single frames, batch, the region frame that stalled
for m in 16 3 4 3 16; do
  export LIBRECTIFY_FLOOD_JIT_FIRST_MAX=$m
  echo "== JIT_FIRST_MAX=$m"
  timeout -k 10 200 python tools/run4k_seeds.py 1 2 3 4 1 2 3 4 2>&1 | python3 -c "
import sys,re
v=[]
for l in sys.stdin:
    m=re.search(r\"'giants_held': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: v.append(float(m.group(5)))
print('   flood ms', v[4:], 'mean %.3f' % (sum(v[4:])/max(len(v[4:]),1)))
"
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   batch %.0f Mpix/s' % d['value'])
"
  for i in 1 2 3; do timeout -k 10 100 python tools/run_regions1080.py 2>&1 | tail -3 | tr "\n" " " | python3 -c "
import sys,re
l=sys.stdin.read()
m=re.search(r\"wall ([\d.]+) ms.*'flood_rounds': (\d+).*'ordered_tail_seeds': (\d+)\", l)
print('   regions 1080p: wall', m.group(1), 'rounds', m.group(2), 'tail', m.group(3))
"; done
done
