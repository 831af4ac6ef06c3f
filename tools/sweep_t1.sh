#!/bin/bash
# On the GPU box: hand-over of walks from the first storage tier to the second -- at so many tiles ("t:N"), or at so many
# tiles when the frontier holds so many records ("w:N:F") -- against the flood time of the natural 4K frame and of the four
# synthetic bench frames.  usage: tools/sweep_t1.sh "t:0 t:128 t:96 t:64 t:48 w:32:6"
for tf in ${1:-t:0 t:128 t:96 t:64 t:48 t:32 w:32:6 w:48:6}; do
  unset LIBRECTIFY_FLOOD_T1_TILES LIBRECTIFY_FLOOD_T1_WIDE_TILES LIBRECTIFY_FLOOD_T1_WIDE_FRONT
  IFS=: read kind a b <<< "$tf"
  if [ $kind = t ]; then export LIBRECTIFY_FLOOD_T1_TILES=$a; else export LIBRECTIFY_FLOOD_T1_WIDE_TILES=$a LIBRECTIFY_FLOOD_T1_WIDE_FRONT=$b; fi
  echo "== $tf"
  python3 tools/run_doc4k.py 2>&1 | tail -1 | sed 's/.*second_tier_seeds/doc4k second_tier_seeds/; s/.slabs.*\[/ [/'
  python3 tools/run4k_seeds.py 2>&1 | sed 's/lines.*second_tier_seeds/second_tier_seeds/; s/.slabs.*\[/ [/'
done
