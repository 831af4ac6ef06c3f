"""The reference's doc image (tests/golden/doc_image_gray.npy, 1000x563 luma) upsampled to 3840x2160 (cubic spline,
scipy.ndimage.zoom) as a natural-image 4K frame: single-frame timing, storage tiers used, optional oracle check."""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import scipy.ndimage as ndi
import librectify_amd as L

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "doc_image_gray.npy")).astype(np.float32) / np.float32(256.0)
W, H = 3840, 2160
img = ndi.zoom(g, (H / g.shape[0], W / g.shape[1]), order=3).astype(np.float32)
img = np.ascontiguousarray(img[:H, :W])
print("frame", img.shape, flush=True)
ctx = L.Context(0)
ctx.set_stage_timing(True)
ctx.set_seed(0)
for rep in range(3):
    t = time.time(); got = ctx.find_line_segment_groups(img, max(W, H) / 100.0); dt = time.time() - t
    print("total %.1f ms" % (dt * 1e3), "lines", len(got), ctx.stage_counters(), ctx.stage_times().round(3), flush=True)
if len(sys.argv) > 1:
    from tests import oracle_lib as O
    ref, _ = O.find_line_segment_groups(img, max(W, H) / 100.0, seed=0)
    a = np.frombuffer(np.ascontiguousarray(got).tobytes(), np.uint32).reshape(len(got), 7)
    b = np.frombuffer(np.ascontiguousarray(ref).tobytes(), np.uint32).reshape(len(ref), 7)
    print("oracle lines", len(ref), "identical", len(got) == len(ref) and bool((a == b).all()))
