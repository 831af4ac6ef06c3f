#!/bin/bash
# logs off / on: bench frames, natural (doc) frame, long bars -- stage times without a profiler
mkdir -p gpurun_out
for m in 0 1; do
  export LIBRECTIFY_FLOOD_LOGS=$m
  echo "== LIBRECTIFY_FLOOD_LOGS=$m"
  timeout -k 10 200 python tools/run4k_seeds.py 2>&1 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"seed (\d+).*'flood_rounds': (\d+).*'log_rewalks': (\d+), 'log_give_ups': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: print('seed', m.group(1), 'flood', m.group(8), 'ms rounds', m.group(2), 'log re-walks', m.group(3), 'sweeps', m.group(4))
" || exit 1
  for f in run_doc4k run_long; do
  timeout -k 10 200 python tools/$f.py 2>&1 | tail -1 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"'flood_rounds': (\d+).*'second_tier_seeds': (\d+).*'log_rewalks': (\d+), 'log_give_ups': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: print('$f', 'flood', m.group(8), 'ms rounds', m.group(1), 'second tier', m.group(2), 'log re-walks', m.group(3), 'sweeps', m.group(4))
    else: print(l)
" || exit 1
  done
done 2>&1 | tee gpurun_out/logs_frames.txt
