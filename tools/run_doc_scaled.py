"""The reference's doc image upsampled to several sizes (cubic spline): how a natural image's flood scales with the
resolution (its walks grow with it: second tier's LDS table, then the teams in global slabs)."""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import scipy.ndimage as ndi
import librectify_amd as L

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "doc_image_gray.npy")).astype(np.float32) / np.float32(256.0)
ctx = L.Context(0)
ctx.set_stage_timing(True)
ctx.set_seed(0)
for W, H in [(1920, 1080), (3840, 2160), (5760, 3240), (7680, 4320)]:
    img = np.ascontiguousarray(ndi.zoom(g, (H / g.shape[0], W / g.shape[1]), order=3).astype(np.float32)[:H, :W])
    for rep in range(3):
        t = time.time(); got = ctx.find_line_segment_groups(img, max(W, H) / 100.0, capacity=200000); dt = time.time() - t
    c = ctx.stage_counters()
    print("%dx%d: %.2f ms (%.0f Mpix/s), flood %.2f ms, %d lines, seeds %d, rounds %d, second tier %d, slabs %d, tail %d" % (W, H, dt * 1e3, W * H / dt / 1e6, ctx.stage_times()[3], len(got), c["seeds"], c["flood_rounds"], c["second_tier_seeds"], c["slabs"], c["ordered_tail_seeds"]), flush=True)
