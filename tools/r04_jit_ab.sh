#!/bin/bash
# rounds just in time off / on: single 4K frames, doc frame, long bars (stage times, no profiler)
for j in 0 1 0 1; do
  export LIBRECTIFY_FLOOD_JIT=$j
  echo "== LIBRECTIFY_FLOOD_JIT=$j"
  timeout -k 10 200 python tools/run4k_seeds.py 2>&1 | python3 -c "
import sys,re
v=[];w=[]
for l in sys.stdin:
    m=re.search(r\"total ([\d.]+) ms.*'log_give_ups': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: v.append(float(m.group(6))); w.append(float(m.group(9)))
print('flood ms', v, 'mean %.3f' % (sum(v)/max(len(v),1)), ' whole frame (stage timers) ms', w, 'mean %.3f' % (sum(w)/max(len(w),1)))
"
  for f in run_doc4k run_long; do
  timeout -k 10 200 python tools/$f.py 2>&1 | tail -1 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"'flood_rounds': (\d+).*\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: print('$f', 'flood', m.group(5), 'ms rounds', m.group(1), 'whole', m.group(8))
"
  done
done
