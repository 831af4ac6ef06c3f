import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle_lib as O
import librectify_amd as L
from librectify_amd import synth
g = np.load("/root/repo/tests/golden/doc_image_gray.npy").astype(np.float32) / np.float32(256.0)
frames = [synth.frame(1920, 1080, 21), synth.frame(960, 540, 4, bars=40), synth.long_bar_frame(1280, 720, 5, K=20),
          synth.region_frame(640, 480, 501), np.ascontiguousarray(g)]
ctx = L.Context(0); ctx.set_seed(0)
for fi, img in enumerate(frames):
    ref = O.find_line_segments(img)
    for mode in (1, 6, 7):
        ctx.set_flood_mode(mode)
        ctx.stage_filter_host(img); ctx.stage_seeds(); ctx.stage_flood()
        c = ctx.stage_counters()
        lab = ctx.download(L.BUF_LABEL)
        print(fi, mode, "ndiff", int((lab != ref["label"]).sum()), "rounds", c["flood_rounds"], "tail", c["ordered_tail_seeds"])
