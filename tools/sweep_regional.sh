#!/bin/bash
# On the GPU box: the earlier hand-over on "regional" frames (LIBRECTIFY_FLOOD_T1_REGIONAL tiles once the frame has sent
# LIBRECTIFY_FLOOD_T1_REGIONAL_MIN walks to the second tier): natural 4K frame and the four synthetic bench frames.
for rm in ${1:-0:16 96:16 64:16 48:16 32:16 64:4 48:4}; do
  export LIBRECTIFY_FLOOD_T1_REGIONAL=${rm%%:*} LIBRECTIFY_FLOOD_T1_REGIONAL_MIN=${rm##*:}
  echo "== regional ${rm%%:*} tiles after ${rm##*:} walks"
  python3 tools/run_doc4k.py 2>&1 | tail -1 | sed 's/.*second_tier_seeds/doc4k second_tier_seeds/; s/.slabs.*\[/ [/'
  python3 tools/run4k_seeds.py 2>&1 | sed 's/lines.*second_tier_seeds/second_tier_seeds/; s/.slabs.*\[/ [/'
done
