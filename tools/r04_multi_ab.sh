#!/bin/bash
# single 4K frames, stage timers on, no profiler: multi-source re-walks off / on, 4 and 8 hardware queues
for q in 4 8; do for m in 0 1; do
  echo "== GPU_MAX_HW_QUEUES=$q LIBRECTIFY_FLOOD_MULTI=$m"
  GPU_MAX_HW_QUEUES=$q LIBRECTIFY_FLOOD_MULTI=$m python tools/run4k_seeds.py 2>&1 | awk '{print $1, $2, $3, $4, $5, $NF, $(NF-1), $(NF-2), $(NF-3), $(NF-4), $(NF-5)}'
done; done
