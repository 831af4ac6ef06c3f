#!/bin/bash
# the batch (headline leg) against the number of blind rounds its lanes enqueue, and the partial-commit limit
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s, %.2f ms per step' % (d['value'], d['ms_per_step']))
"
}
echo "== default (previous frame's rounds + 2, at least 6)"; run
echo "== BLIND_EXTRA=1 BLIND_MIN=5"; LIBRECTIFY_BLIND_EXTRA=1 LIBRECTIFY_BLIND_MIN=5 run
echo "== BLIND_EXTRA=3 BLIND_MIN=8"; LIBRECTIFY_BLIND_EXTRA=3 LIBRECTIFY_BLIND_MIN=8 run
echo "== PARTIAL_ROUNDS=3"; LIBRECTIFY_FLOOD_PARTIAL_ROUNDS=3 run
echo "== PARTIAL_ROUNDS=2"; LIBRECTIFY_FLOOD_PARTIAL_ROUNDS=2 run
echo "== default"; run
