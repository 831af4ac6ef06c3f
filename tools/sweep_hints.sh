#!/bin/bash
# On the GPU box: hold-back from the start (forced on / off) x early hand-over on regional frames (never / after 16 / after 1
# wide walks): flood ms of the long-edge stress frame, the natural 4K frame and synthetic frames 3 and 4.
for hh in 0 1; do for rm in 1000000 16 4 1; do
  export LIBRECTIFY_FLOOD_HOLD_HINT=$hh LIBRECTIFY_FLOOD_T1_REGIONAL_MIN=$rm
  echo "== hold from start $hh, regional after $rm wide walks"
  python3 tools/run_long.py 2>&1 | tail -1 | sed 's/.*second_tier_seeds/long  second_tier_seeds/; s/.slabs.*\[/ [/'
  python3 tools/run_doc4k.py 2>&1 | tail -1 | sed 's/.*second_tier_seeds/doc4k second_tier_seeds/; s/.slabs.*\[/ [/'
  python3 tools/run4k_seeds.py 3 4 2>&1 | sed 's/lines.*second_tier_seeds/second_tier_seeds/; s/.slabs.*\[/ [/'
done; done
