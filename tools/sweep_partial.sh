#!/bin/bash
# On the GPU box: flood time of the four bench frames and the batch rate for variants of the partial commits.
run() {
  echo "== $1"
  env $2 python3 tools/run4k_seeds.py 2>&1 | sed -E "s/.*flood_rounds': ([0-9]+).*walked_px': ([0-9]+).*\[ *[0-9.]+ +[0-9.]+ +[0-9.]+ +([0-9.]+).*/rounds \1 walked \2 flood_ms \3/"
  env $2 python3 bench.py --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "import sys,json; b=json.loads(sys.stdin.read()); print('batch', b['value'], 'device-resident', b['other_rates_Mpix_per_s'])"
}
run "off" "LIBRECTIFY_FLOOD_PARTIAL=0"
run "on, 8 steps" "LIBRECTIFY_FLOOD_PARTIAL_STEPS=8"
run "on, 12 steps" "LIBRECTIFY_FLOOD_PARTIAL_STEPS=12"
run "on, 16 steps" "LIBRECTIFY_FLOOD_PARTIAL_STEPS=16"
run "on, 24 steps" "LIBRECTIFY_FLOOD_PARTIAL_STEPS=24"
run "off again" "LIBRECTIFY_FLOOD_PARTIAL=0"
