#!/bin/bash
# On the GPU box: per-walk early hand-over (at N tiles when the frontier holds F records: "N:F") against the flood time of
# the long-bar frame, the natural 4K frame and the four synthetic bench frames.  usage: tools/sweep_wide.sh "0:0 96:8 64:8 96:6"
for tf in ${1:-0:0 128:8 96:8 64:8 96:6 64:6 96:12 64:12}; do
  unset LIBRECTIFY_FLOOD_T1_WIDE_TILES LIBRECTIFY_FLOOD_T1_WIDE_FRONT
  if [ ${tf%%:*} != 0 ]; then export LIBRECTIFY_FLOOD_T1_WIDE_TILES=${tf%%:*} LIBRECTIFY_FLOOD_T1_WIDE_FRONT=${tf##*:}; fi
  echo "== wide hand-over $tf"
  python3 tools/run_long.py 2>&1 | tail -1 | sed 's/.*second_tier_seeds/long  second_tier_seeds/; s/.slabs.*\[/ [/'
  python3 tools/run_doc4k.py 2>&1 | tail -1 | sed 's/.*second_tier_seeds/doc4k second_tier_seeds/; s/.slabs.*\[/ [/'
  python3 tools/run4k_seeds.py 2>&1 | sed 's/lines.*second_tier_seeds/second_tier_seeds/; s/.slabs.*\[/ [/'
done
