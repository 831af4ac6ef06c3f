#!/bin/bash
# On the GPU box: early hand-over of wide walks from the first storage tier to the second (tiles walked, frontier records)
# against the flood time of the natural 4K frame and of the four synthetic bench frames.
# usage: tools/sweep_t1.sh "0:6 32:4 32:6 32:8 48:4 48:6 64:6"
for tf in ${1:-0:6 32:4 32:6 32:8 48:4 48:6 64:6}; do
  export LIBRECTIFY_FLOOD_T1_WIDE_TILES=${tf%%:*} LIBRECTIFY_FLOOD_T1_WIDE_FRONT=${tf##*:}
  echo "== wide tiles ${tf%%:*} front ${tf##*:}"
  python3 tools/run_doc4k.py 2>&1 | tail -1 | sed 's/.*second_tier_seeds/doc4k second_tier_seeds/; s/.slabs.*\[/ [/'
  python3 tools/run4k_seeds.py 2>&1 | sed 's/lines.*second_tier_seeds/second_tier_seeds/; s/.slabs.*\[/ [/'
done
