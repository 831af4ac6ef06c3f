#!/bin/bash
for s in 5 6 7 8 5; do
  echo "== resident frames, $s lanes"
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs --host-memory device --streams $s 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s' % d['value'])
"
done
