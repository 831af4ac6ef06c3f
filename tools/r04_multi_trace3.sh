#!/bin/bash
# timelines of single 4K frames (seeds 1, 4) at several way-point thresholds: flood span and per-round explore / multi-team times
mkdir -p gpurun_out
for cfg in "$@"; do
  for seed in 1 4; do
    if [ $cfg = off ]; then export LIBRECTIFY_FLOOD_MULTI=0; unset LIBRECTIFY_FLOOD_MULTI_MIN; else export LIBRECTIFY_FLOOD_MULTI=1; export LIBRECTIFY_FLOOD_MULTI_MIN=$cfg; fi
    bash tools/single_frame_trace.sh gpurun_out/sft_m${cfg}_s$seed $seed > /dev/null 2>&1
    python3 - gpurun_out/sft_m${cfg}_s$seed $cfg $seed <<'PY'
import csv, re, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(d + '/kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = max(i for i, r in enumerate(rows) if 'filter_lanes' in r['Kernel_Name'])
fl = [r for r in rows[last:] if 'flood_' in r['Kernel_Name']]
t0 = int(fl[0]['Start_Timestamp'])
# the flood ends with the last kernel that ran longer than an empty launch
real = [r for r in fl if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 6000]
span = (int(real[-1]['End_Timestamp']) - t0) / 1e3
rounds, cur = [], None
for r in fl:
    n = re.sub(r'.*flood_', '', r['Kernel_Name']).split('(')[0].replace('_kernel', '')
    if n == 'survivors' and cur is not None:
        rounds.append((int(r['End_Timestamp']) - cur) / 1e3); cur = None
    elif cur is None and n in ('explore', 'explore_team'):
        cur = int(r['Start_Timestamp'])
print("multi %s seed %s: flood to its last working kernel %.0f us; rounds %s" % (sys.argv[2], sys.argv[3], span, ' '.join('%.0f' % x for x in rounds[:7])), open(d + '/run.txt').read().strip().split('\n')[-1].split('multi_source_walks')[-1])
PY
  done
done
