#!/bin/bash
# the hold-back line (per cent of the seed order that keeps walking) now that long late lists are harmless: natural frame, long bars, regions, bench frames
for h in 80 70 60 50 90; do
  export LIBRECTIFY_FLOOD_HOLD=$h
  echo "== LIBRECTIFY_FLOOD_HOLD=$h"
  for f in run_doc4k run_long run_edgeless; do timeout -k 10 100 python3 tools/$f.py 2>&1 | tail -1 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"total ([\d.]+) ms.*'flood_rounds': (\d+).*'giants_held': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: print('   $f: total', m.group(1), 'flood', m.group(7), 'ms rounds', m.group(2))
"; done
  timeout -k 10 100 python tools/run_regions1080.py 2>&1 | tail -3 | tr "\n" " " | python3 -c "
import sys,re
l=sys.stdin.read()
m=re.search(r\"wall ([\d.]+) ms.*'flood_rounds': (\d+)\", l)
print('   regions 1080p: wall', m.group(1), 'rounds', m.group(2))
"
  timeout -k 10 200 python tools/run4k_seeds.py 2>&1 | python3 -c "
import sys,re
v=[]
for l in sys.stdin:
    m=re.search(r\"'giants_held': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: v.append(float(m.group(5)))
print('   bench frames flood ms', v)
"
done
