"""One 8192x8192 frame through find_line_segment_groups from a pageable and from a page-locked buffer: where the
upload's time goes at BASELINE config 5's size (LIBRECTIFY_CALL_DEBUG=1 prints the host's view of each call)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import librectify_amd as L
from librectify_amd import synth

W = H = 8192
img = synth.frame(W, H, 7, bars=6000, tile=512)
ctx = L.Context(0)
pinned = ctx.host_alloc((H, W))
pinned[:] = img
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, f in (("pageable", img), ("page-locked", pinned)):
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        lines = ctx.find_line_segment_groups(f, 20.0, num_threads=nt, capacity=200000)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s, num_threads %d: %s ms" % (name, nt, np.round(ts, 2)), flush=True)
