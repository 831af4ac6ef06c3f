#!/bin/bash
# BASELINE config 4 (512 x 1920x1080 from pageable host pointers) against the number of lanes
for s in 5 7 9 12 5; do
  echo "== $s lanes"
  timeout -k 10 300 python bench.py --config batch1080 --streams $s --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   %.1f ms per pass of 512 frames, value %.0f %s' % (d['ms_per_step'], d['value'], d['unit']))
"
done
