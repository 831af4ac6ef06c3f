#!/bin/bash
# On the GPU box: batch rate against the number of lanes (frames in flight), lanes spinning or sleeping in their wait,
# for pageable / page-locked / HBM-resident frames.  usage: tools/sweep_lanes.sh "6 8 10 12"
for s in ${1:-6 8 10 12}; do
  for sl in 0 1; do
    if [ $sl = 1 ]; then export LIBRECTIFY_LANES_SLEEP=1; else unset LIBRECTIFY_LANES_SLEEP; fi
    GPU_MAX_HW_QUEUES=$((s > 8 ? 16 : 8)) python3 bench.py --streams $s --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
b = json.loads(sys.stdin.read())
o = b['other_rates_Mpix_per_s']
print('lanes $s sleep $sl: pageable %.0f  page-locked %.0f  HBM-resident %.0f Mpix/s' % (b['value'], o['host_pinned'], o['device_resident']))"
  done
done
