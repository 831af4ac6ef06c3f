"""One-off soak check: label images and records of many random frames (sizes, bar counts, noise) against the oracle,
through single calls and the staged API.  Not part of the test suite (it takes minutes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
import librectify_amd as L
from librectify_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.RandomState(2026)
ctx = L.Context(0)
ctx.set_seed(0)
bad = 0
t0 = time.time()
for i in range(n):
    w = int(rng.randint(64, 1400)); h = int(rng.randint(64, 1000))
    bars = int(rng.randint(2, 120))
    img = synth.frame(w, h, 5000 + i, bars=bars, noise=float(rng.choice([0.0, 0.002, 0.005, 0.02])))
    if rng.rand() < 0.3:  # long edges: walks that outgrow the storage tiers
        img[:, : w // 2] += np.linspace(0, 0.3, h, dtype=np.float32)[:, None]
    ref = O.find_line_segments(img)
    ml = max(w, h) / 100.0
    full, _ = O.find_line_segment_groups(img, ml, seed=0)
    ctx.set_flood_mode(int(rng.choice([1, 1, 1, 2, 3, 5, 6, 7])))
    ctx.stage_filter_host(img); ctx.stage_seeds(); ctx.stage_flood()
    lab = ctx.download(L.BUF_LABEL)
    ok = np.array_equal(lab, ref["label"])
    got = ctx.find_line_segment_groups(img, ml)
    ok = ok and got.tobytes() == full.tobytes()
    if not ok:
        bad += 1
        print("MISMATCH frame", i, w, h, bars, flush=True)
    if i % 25 == 0:
        print(i, "frames, %.0f s" % (time.time() - t0), ctx.stage_counters(), flush=True)
print("done:", n, "frames,", bad, "mismatches")
sys.exit(1 if bad else 0)
