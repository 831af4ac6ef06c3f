#!/bin/bash
# thresholds of the logs again (with the partial-commit limit and the rounds just in time): single 4K frames, doc frame
for cfg in "16 12" "16 16" "24 16" "24 24" "32 16" "16 12"; do
  set -- $cfg
  echo "== LOG_MIN=$1 LOG_WALK=$2"
  export LIBRECTIFY_FLOOD_LOG_MIN=$1 LIBRECTIFY_FLOOD_LOG_WALK=$2
  timeout -k 10 200 python tools/run4k_seeds.py 1 2 3 4 1 2 3 4 2>&1 | python3 -c "
import sys,re
v=[]
for l in sys.stdin:
    m=re.search(r\"'log_give_ups': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: v.append(float(m.group(5)))
print('   flood ms', v[4:], 'mean %.3f' % (sum(v[4:])/max(len(v[4:]),1)))
"
  timeout -k 10 200 python tools/run_doc4k.py 2>&1 | tail -1 | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r\"'flood_rounds': (\d+).*\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: print('   doc flood', m.group(5), 'ms rounds', m.group(1))
"
done
