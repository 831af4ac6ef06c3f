"""Per-round statistics of the flood on a frame (LIBRECTIFY_FLOOD_DEBUG=1 prints them; rounds are then synchronised
one at a time).  usage: flood_debug.py [bench|doc|long|1080]"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["LIBRECTIFY_FLOOD_DEBUG"] = "1"
import numpy as np
import librectify_amd as L
from librectify_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "bench"
if which == "bench":
    img = synth.frame(3840, 2160, 1)
elif which.startswith("seed"):
    img = synth.frame(3840, 2160, int(which[4:]))
elif which == "1080":
    img = synth.frame(1920, 1080, 1000)
elif which == "doc":
    import scipy.ndimage as ndi
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "doc_image_gray.npy")).astype(np.float32) / np.float32(256.0)
    img = np.ascontiguousarray(ndi.zoom(g, (2160 / g.shape[0], 3840 / g.shape[1]), order=3).astype(np.float32)[:2160, :3840])
ctx = L.Context(0)
ctx.set_stage_timing(True)
h, w = img.shape
for rep in range(2):
    sys.stderr.write("---- pass %d\n" % rep)
    ctx.find_line_segment_groups(img, max(w, h) / 100.0)
print(ctx.stage_counters(), ctx.stage_times().round(3))
