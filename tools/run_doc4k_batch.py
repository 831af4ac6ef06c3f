"""Batch throughput on the natural 4K frame of tools/run_doc4k.py: frames resident in HBM, then the same frames from
pageable host memory (upload inside the call).  usage: tools/run_doc4k_batch.py [lanes]"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, time
import scipy.ndimage as ndi
import librectify_amd as L

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "doc_image_gray.npy")).astype(np.float32) / np.float32(256.0)
W, H = 3840, 2160
base = np.ascontiguousarray(ndi.zoom(g, (H / g.shape[0], W / g.shape[1]), order=3).astype(np.float32)[:H, :W])
rng = np.random.RandomState(0)
B = 32
frames = np.stack([base + rng.normal(0, 0.002, base.shape).astype(np.float32) for _ in range(4)] * (B // 4))
ctx = L.Context(0)
ctx.set_seed(0)
ctx.set_batch_streams(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
d = ctx.device_upload(frames)
for rep in range(4):
    t = time.time()
    out, n, tf = ctx.find_line_segment_groups_batch_device(d, W * H, B, W, H, max(W, H) / 100.0, capacity=4096)
    dt = time.time() - t
    print("%d frames: %.1f ms, %.0f Mpix/s, mean segments %.0f" % (B, dt * 1e3, B * W * H / dt / 1e6, float(np.mean(n))), flush=True)
ctx.device_free(d)
for rep in range(4):
    t = time.time()
    out, n, tf = ctx.find_line_segment_groups_batch_host(frames, max(W, H) / 100.0, num_threads=12, capacity=4096)
    dt = time.time() - t
    print("%d frames from pageable host memory: %.1f ms, %.0f Mpix/s, mean segments %.0f" % (B, dt * 1e3, B * W * H / dt / 1e6, float(np.mean(n))), flush=True)
