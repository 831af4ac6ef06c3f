"""Does a host-to-device DMA running next to the kernels slow them down?  The batch entry on frames that are already in
HBM (no upload of its own), timed alone and with a background thread that copies 33 MB blocks from page-locked host
memory to a scratch buffer on its own stream, at full rate or throttled to about one block per frame."""
import os, sys, time, threading; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import librectify_amd as L
from librectify_amd import synth

W, H, B = 3840, 2160, 64
ctx = L.Context(0)
ctx.set_batch_streams(6)
bases = [synth.frame(W, H, s) for s in (1, 2, 3, 4)]
frames = np.stack([bases[i % 4] if (i // 4) % 2 == 0 else bases[i % 4][:, ::-1] for i in range(B)]).astype(np.float32)
d = torch.from_numpy(frames).cuda()
out = np.zeros((B, 8192), L.LINE_DTYPE)
def step():
    ctx.find_line_segment_groups_batch_device(d.data_ptr(), H * W, B, W, H, float(W) / 100.0, capacity=8192, cfg=None, out=out)
def rate(n=4):
    step()
    t = time.time()
    for _ in range(n): step()
    return n * B * W * H / (time.time() - t) / 1e6

src = torch.empty((H, W), dtype=torch.float32).pin_memory()
dst = torch.empty((4, H, W), dtype=torch.float32, device="cuda")
stop = False
count = [0]
def pump(gap_s):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        i = 0
        while not stop:
            dst[i & 3].copy_(src, non_blocking=True)
            s.synchronize()
            count[0] += 1
            i += 1
            if gap_s: time.sleep(gap_s)

print("alone: %.0f Mpix/s" % rate(), flush=True)
for gap, name in ((0.0, "DMA back to back"), (0.0004, "DMA about one frame per 1.1 ms")):
    stop = False; count[0] = 0
    th = threading.Thread(target=pump, args=(gap,)); th.start()
    time.sleep(0.05)
    t0 = time.time(); c0 = count[0]
    r = rate()
    dt = time.time() - t0; n = count[0] - c0
    stop = True; th.join()
    print("%s: %.0f Mpix/s  (%.1f GB/s of uploads meanwhile)" % (name, r, n * H * W * 4 / dt / 1e9), flush=True)
print("alone again: %.0f Mpix/s" % rate(), flush=True)
