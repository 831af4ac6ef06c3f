#!/bin/bash
# what the lanes and the uploader of a batch call wait for (LIBRECTIFY_LANE_DEBUG), pageable / pinned / device-resident frames
mkdir -p gpurun_out
for kind in pageable pinned; do
  LIBRECTIFY_LANE_DEBUG=1 python bench.py --steps 2 --warmup 1 --no-extra-legs --no-cpu-baseline --host-memory $kind > gpurun_out/r04_lane_$kind.json 2> gpurun_out/r04_lane_$kind.err
  python3 - gpurun_out/r04_lane_$kind.err $kind <<'PY'
import re, sys, numpy as np
lane, up = [], []
for l in open(sys.argv[1]):
    m = re.search(r"lane (\d+) frame (\d+): enqueue ([\d.]+) ms, wait ([\d.]+), whole call ([\d.]+); device total ([\d.]+); upload done (-?[\d.]+) ms", l)
    if m: lane.append([float(x) for x in m.groups()])
    m = re.search(r"uploader frame (\d+): slot (\d+), waited (\d+) naps for it, staged and enqueued in ([\d.]+) ms, on the link ([\d.]+) ms", l)
    if m: up.append([float(x) for x in m.groups()])
lane, up = np.array(lane), np.array(up)
n = len(lane) // 3 * 2  # (skip the warm-up step)
L, U = lane[-n:], up[-n:]
print("%s: per frame on a lane: enqueue %.3f ms, wait %.3f, whole call %.3f, device total %.3f; upload finished %.2f ms (median) before its first kernel, late (<0.05 ms lead) for %d of %d frames"
      % (sys.argv[2], L[:,2].mean(), L[:,3].mean(), L[:,4].mean(), L[:,5].mean(), np.median(L[:,6]), int((L[:,6] < 0.05).sum()), len(L)))
print("   uploader: naps for a free slot %.1f per frame (frames that had to nap: %d of %d), staged + enqueued in %.3f ms, on the link %.3f ms" % (U[:,2].mean(), int((U[:,2] > 0).sum()), len(U), U[:,3].mean(), U[:,4].mean()))
PY
done
