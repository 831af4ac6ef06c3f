#!/bin/bash
# On the GPU box: the hold-back line (percent of the seeds that do not wait; 0 = no hold-back) against the flood time of
# the natural 4K frame and of the four synthetic bench frames.
for h in ${1:-80 0 90 70 60}; do
  export LIBRECTIFY_FLOOD_HOLD=$h
  echo "== hold-back line at $h %"
  python3 tools/run_doc4k.py 2>&1 | tail -1 | sed 's/.*flood_rounds/doc4k flood_rounds/; s/.labelled.*\[/ [/'
  python3 tools/run4k_seeds.py 2>&1 | sed 's/lines.*flood_rounds/flood_rounds/; s/.labelled.*\[/ [/'
done
