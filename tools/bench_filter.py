"""Filter-stage micro-benchmark: the fused filter kernel alone on N distinct device-resident
frames (enough bytes per lap to defeat the 256 MiB Infinity Cache), timed with the library's HIP events."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import librectify_amd as L

W = int(os.environ.get("W", 3840)); H = int(os.environ.get("H", 2160))
N = int(os.environ.get("NFRAMES", 12)); LAPS = int(os.environ.get("LAPS", 5))
# contexts in turn: each has a workspace of its own, so the kernel's OUTPUT (9 B/px: dx, dy, mask) lands in other memory
# every launch as well -- four 4K workspaces are 300 MB of outputs, more than the 256 MiB Infinity Cache holds
NCTX = int(os.environ.get("NCTX", 4 if W * H < 32 * 1024 * 1024 else 1))
rng = np.random.RandomState(0)
d = torch.empty((N, H, W), dtype=torch.float32, device="cuda")
base = (rng.rand(H, W) * 0.1 + 0.45).astype(np.float32)
for i in range(N):
    d[i].copy_(torch.from_numpy(np.roll(base, 31 * i, axis=1)))
torch.cuda.synchronize()
ctxs = [L.Context(0) for _ in range(NCTX)]
ts = []
for lap in range(LAPS):
    for i in range(N):
        ctx = ctxs[(lap * N + i) % NCTX]
        ctx.stage_filter_device(d.data_ptr() + i * H * W * 4, W, H)
        ctx.synchronize()
        if lap > 0:
            ts.append(float(ctx.stage_times_partial()))
ts = np.array(ts)
print("(%d input frames, %d workspaces in turn)" % (N, NCTX))
print("filter kernel %dx%d: mean %.2f us  min %.2f us  -> %.0f GB/s algorithmic (18 B/px), frac of 8 TB/s %.3f" % (
    W, H, ts.mean() * 1e3, ts.min() * 1e3, 18.0 * W * H / (ts.mean() * 1e-3) / 1e9, 18.0 * W * H / (ts.mean() * 1e-3) / 8e12))
