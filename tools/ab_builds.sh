#!/bin/bash
# A/B of two builds of the library on the GPU box: bench.py's pinned-host throughput and the device-resident leg
for rep in 1 2 3; do
  for v in main tm; do
    cp alt/$v.so librectify_amd/librectify_amd.so
    echo "== $v rep $rep"
    GPU_MAX_HW_QUEUES=8 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --host-memory pinned 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['other_rates_Mpix_per_s'].get('device_resident'), d['roofline']['kernel_ms'], d['stage_ms_per_frame'])"
  done
done
