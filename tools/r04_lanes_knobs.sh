#!/bin/bash
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s' % d['value'])
"
}
for rep in 1 2; do
echo "== as it is"; run
for p in 1 2 4; do echo "== PARTIAL_ROUNDS=$p"; LIBRECTIFY_FLOOD_PARTIAL_ROUNDS=$p run; done
for j in 5 50; do echo "== JIT_LANES=$j"; LIBRECTIFY_FLOOD_JIT_LANES=$j run; done
done
