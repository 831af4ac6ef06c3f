"""Two named frames (tools/frames.py) in turn on ONE context: what a frame costs when the context's hints come from the other
kind of content.  usage: alternate_frames.py <name a> <name b> [turns]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import frames
import numpy as np
import librectify_amd as L

a, b = sys.argv[1], sys.argv[2]
turns = int(sys.argv[3]) if len(sys.argv) > 3 else 5
fa, fb = frames.make(a), frames.make(b)
ctx = L.Context(0)
ctx.set_stage_timing(True)
res = {a: [], b: []}
fl = {a: [], b: []}
for t in range(turns):
    for name, (img, ml) in ((a, fa), (b, fb)):
        t0 = time.perf_counter()
        ctx.find_line_segment_groups(img, ml)
        res[name].append((time.perf_counter() - t0) * 1e3)
        fl[name].append(float(ctx.stage_times()[L.T_FLOOD]))
for name in (a, b):
    print("%-14s wall %s ms | flood %s" % (name, " ".join("%.2f" % d for d in res[name]), " ".join("%.2f" % d for d in fl[name])))
# the same frames, each kind in a row
for name, (img, ml) in ((a, fa), (b, fb)):
    d = []
    for t in range(4):
        t0 = time.perf_counter(); ctx.find_line_segment_groups(img, ml); d.append((time.perf_counter() - t0) * 1e3)
    print("%-14s in a row: %s ms" % (name, " ".join("%.2f" % x for x in d)))
