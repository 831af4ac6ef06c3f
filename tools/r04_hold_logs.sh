#!/bin/bash
# the hold-back of the weakest seeds with the logs on: natural frame, long bars, regions
for h in -1 100; do
  if [ $h = -1 ]; then unset LIBRECTIFY_FLOOD_HOLD; echo "== hold-back as it is (80 %)"; else export LIBRECTIFY_FLOOD_HOLD=$h; echo "== LIBRECTIFY_FLOOD_HOLD=$h (none)"; fi
  for f in run_doc4k run_long; do
  timeout -k 10 200 python tools/$f.py 2>&1 | tail -2 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"'flood_rounds': (\d+).*'second_tier_seeds': (\d+).*'log_rewalks': (\d+), 'log_give_ups': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: print('   $f', 'flood', m.group(8), 'ms rounds', m.group(1), 'second tier', m.group(2), 'log re-walks', m.group(3), 'sweeps', m.group(4))
"
  done
  timeout -k 10 200 python tools/time_regions.py 2>&1 | tail -3
done
