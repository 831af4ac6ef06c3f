#!/bin/bash
# thresholds of the logs (tiles a walk must have to leave one / tiles a seed with a log walks first), single 4K frames, then bench.py off / on
mkdir -p gpurun_out
export LIBRECTIFY_FLOOD_LOGS=1
for cfg in "12 12" "8 8" "6 6" "16 12" "8 12" "12 8"; do
  set -- $cfg
  echo "== LOG_MIN=$1 LOG_WALK=$2"
  LIBRECTIFY_FLOOD_LOG_MIN=$1 LIBRECTIFY_FLOOD_LOG_WALK=$2 timeout -k 10 200 python tools/run4k_seeds.py 2>&1 | python3 -c "
import sys,re
v=[]
for l in sys.stdin:
    m=re.search(r\"'log_rewalks': (\d+), 'log_give_ups': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: v.append(float(m.group(6)))
print('flood ms', v, 'mean %.3f' % (sum(v)/max(len(v),1)))
" || exit 1
done 2>&1 | tee gpurun_out/logs_sweep.txt
for m in 0 1; do
  echo "== bench.py LIBRECTIFY_FLOOD_LOGS=$m"
  LIBRECTIFY_FLOOD_LOGS=$m timeout -k 10 500 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'ms_per_step')}, {k: d.get(k) for k in ('flood', 'single_frame', 'resident', 'natural_frame') if k in d})
"
done 2>&1 | tee -a gpurun_out/logs_sweep.txt
