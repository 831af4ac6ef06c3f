#!/bin/bash
# at how many tiles a second-tier walk counts as a giant (the team's table holds 1536): region frames, natural frame, long bars, bench frames
for t in 0 1024 640 384; do
  export LIBRECTIFY_FLOOD_TEAM_TILES=$t
  echo "== TEAM_TILES=$t (0: the table's 1536)"
  timeout -k 10 100 python tools/run_regions1080.py 2>&1 | tail -3 | tr "\n" " " | python3 -c "
import sys,re
l=sys.stdin.read()
m=re.search(r\"wall ([\d.]+) ms.*'flood_rounds': (\d+).*'ordered_tail_seeds': (\d+).*'giants_held': (\d+)\", l)
print('   regions 1080p: wall', m.group(1), 'rounds', m.group(2), 'tail', m.group(3), 'held', m.group(4))
"
  timeout -k 10 100 python3 tools/time_regions.py 2>&1 | tail -10 | grep -E "frame [01346] " | cut -c1-105
  for f in run_doc4k run_long run_edgeless; do timeout -k 10 100 python3 tools/$f.py 2>&1 | tail -1 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"total ([\d.]+) ms.*'flood_rounds': (\d+).*'giants_held': (\d+)\", l)
    if m: print('   $f: total', m.group(1), 'ms rounds', m.group(2), 'held', m.group(3))
"; done
  timeout -k 10 200 python tools/run4k_seeds.py 2>&1 | python3 -c "
import sys,re
v=[]
for l in sys.stdin:
    m=re.search(r\"'giants_held': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: v.append(float(m.group(5)))
print('   bench frames flood ms', v)
"
done
