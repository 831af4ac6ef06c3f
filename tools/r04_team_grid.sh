#!/bin/bash
# grid of the second tier's launch: the batch (headline leg), single frames
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s, %.2f ms per step' % (d['value'], d['ms_per_step']))
"
}
for g in 512 32 512 32 8 128; do echo "== TEAM_GRID=$g"; LIBRECTIFY_FLOOD_TEAM_GRID=$g run; done
echo "== by hint"; run
for g in 512 32; do echo "== single frames TEAM_GRID=$g"; LIBRECTIFY_FLOOD_TEAM_GRID=$g timeout -k 10 200 python tools/run4k_seeds.py 2>&1 | python3 -c "
import sys,re
v=[]
for l in sys.stdin:
    m=re.search(r\"'giants_held': (\d+)\} \[\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\", l)
    if m: v.append(float(m.group(5)))
print('   flood ms', v)
"; done
