#!/bin/bash
# On the GPU box: A/B of two builds of the library kept as alt/main.so and alt/tm.so (untracked): bench.py's pinned-host
# throughput, the device-resident leg, per-round explore times of a single frame, alternating.  usage: tools/ab_builds.sh [reps]
for rep in $(seq 1 ${1:-2}); do
  for v in main tm; do
    cp alt/$v.so librectify_amd/librectify_amd.so
    echo "== $v rep $rep"
    GPU_MAX_HW_QUEUES=8 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --host-memory pinned 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['other_rates_Mpix_per_s'].get('device_resident'), d['roofline']['kernel_ms'], d['stage_ms_per_frame'])"
    python3 tools/flood_debug.py bench 2>&1 | grep "explore kernels" | tail -6 | tr '\n' ' '; echo
  done
done
