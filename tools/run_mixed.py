"""Natural and synthetic 4K frames alternating on one context: what the hints carried from frame to frame (second tier,
hold-back, early hand-over) cost when the next frame is of the other kind."""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import scipy.ndimage as ndi
import librectify_amd as L
from librectify_amd import synth

W, H = 3840, 2160
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "doc_image_gray.npy")).astype(np.float32) / np.float32(256.0)
doc = np.ascontiguousarray(ndi.zoom(g, (H / g.shape[0], W / g.shape[1]), order=3).astype(np.float32)[:H, :W])
syn = synth.frame(W, H, 1)
ctx = L.Context(0)
ctx.set_stage_timing(True)
ctx.set_seed(0)
for name, img in [("doc", doc), ("doc", doc), ("doc", doc), ("syn", syn), ("syn", syn), ("syn", syn), ("doc", doc), ("syn", syn), ("doc", doc), ("doc", doc)]:
    t = time.time(); got = ctx.find_line_segment_groups(img, max(W, H) / 100.0); dt = time.time() - t
    c = ctx.stage_counters()
    print("%s total %.1f ms, flood %.3f ms, rounds %d, second tier %d, laps %d" % (name, dt * 1e3, ctx.stage_times()[3], c["flood_rounds"], c["second_tier_seeds"], c["frame_laps"]), flush=True)
