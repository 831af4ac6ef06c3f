"""One 4K frame at a time through the reference's own entry (find_line_segment_groups + compute_rectification_transform):
wall time from pageable and page-locked buffers, four bench frames.  Env knobs of the upload path are read by the
library (LIBRECTIFY_UPLOAD_BAND_KB, LIBRECTIFY_FILTER_EVERY)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import librectify_amd as L
from librectify_amd import synth

W, H = 3840, 2160
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = [synth.frame(W, H, s) for s in (1, 2, 3, 4)]
ctx = L.Context(0)
pinned = ctx.host_alloc((4, H, W))
for i in range(4):
    pinned[i] = frames[i]
for name, src in (("pageable", frames), ("page-locked", pinned)):
    per = []
    for f in src:
        ts = []
        for rep in range(6):
            t0 = time.perf_counter()
            lines = L.find_line_segment_groups(f, 38.4, num_threads=nt)
            L.compute_rectification_transform(lines, W, H)
            ts.append(time.perf_counter() - t0)
        per.append(np.mean(ts[1:]) * 1e3)
    print("%s, num_threads %d: %s ms per frame, mean %.3f ms = %.0f Mpix/s" % (name, nt, np.round(per, 3), np.mean(per), W * H / np.mean(per) / 1e3), flush=True)
