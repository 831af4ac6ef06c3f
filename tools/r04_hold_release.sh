#!/bin/bash
# when the hold-back ends (active seeds left below the line), logs on: natural frame, long bars, regions
for r in 64 256 1024 4096 16384; do
  export LIBRECTIFY_FLOOD_HOLD_RELEASE=$r; echo "== LIBRECTIFY_FLOOD_HOLD_RELEASE=$r"
  for f in run_doc4k run_long; do
  timeout -k 10 200 python tools/$f.py 2>&1 | tail -1 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"'flood_rounds': (\d+).*'second_tier_seeds': (\d+).*'log_rewalks': (\d+), 'log_give_ups': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: print('   $f', 'flood', m.group(8), 'ms rounds', m.group(1), 'second tier', m.group(2))
"
  done
  timeout -k 10 200 python tools/time_regions.py 2>&1 | tail -3 | cut -c1-60
done
