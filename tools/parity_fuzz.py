"""Random frames of every kind this repository knows, in random order on ONE context (so that every hint a context hands from
frame to frame meets every other kind of content), sizes incl. ragged ones: label image and records against the oracle after the
staged calls, the grouped lines of the full call as well.  usage: parity_fuzz.py [n_frames] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import librectify_amd as L
from librectify_amd import synth

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
sizes = [(640, 480), (960, 540), (1283, 717), (1280, 720), (801, 603), (1920, 1080)]


def make(kind, w, h, s):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    if kind == "bars":
        return synth.frame(w, h, s, bars=int(rng.randint(10, 80)))
    if kind == "regions":
        return synth.region_frame(w, h, 400 + s)
    if kind == "ramp":
        return synth.ramp_frame(w, h, s)
    if kind == "longbars":
        return synth.long_bar_frame(w, h, s, K=int(rng.randint(6, 24)))
    if kind == "radial":
        return (1.0 - np.hypot(xx - w / 2, yy - h / 2) / np.hypot(w / 2, h / 2)).astype(np.float32)
    if kind == "stripes":
        p = int(rng.choice([6, 17, 40]))
        return (0.5 + 0.4 * np.sin((xx + 0.5 * yy) * 2 * np.pi / p)).astype(np.float32)
    if kind == "noise":
        return (rng.rand(h, w) * float(rng.choice([0.05, 0.3, 1.0]))).astype(np.float32)
    if kind == "radial_noise":
        return ((1.0 - np.hypot(xx - w / 2, yy - h / 2) / np.hypot(w / 2, h / 2)) + rng.rand(h, w) * 0.02).astype(np.float32)
    if kind == "constant":
        return np.full((h, w), 0.5, np.float32)
    raise ValueError(kind)


kinds = ["bars", "regions", "ramp", "longbars", "radial", "stripes", "noise", "radial_noise", "constant"]
ctx = L.Context(0)
ctx.set_seed(0)
bad = 0
t_all = time.perf_counter()
for i in range(n_frames):
    kind = kinds[int(rng.randint(len(kinds)))]
    w, h = sizes[int(rng.randint(len(sizes)))] if rng.rand() < 0.6 else sizes[i % 2]  # (runs of one size: the hints survive)
    img = np.ascontiguousarray(make(kind, w, h, int(rng.randint(1, 1000))))
    ref = O.find_line_segments(img, num_threads=16)
    ctx.stage_filter_host(img)
    ctx.stage_seeds()
    ctx.stage_flood()
    ok_l = bool((ctx.download(L.BUF_LABEL) == ref["label"]).all())
    c = ctx.stage_counters()
    ok_r = ctx.stage_fit().tobytes() == ref["lines"].tobytes()
    ml = float(max(w, h)) / 100.0
    ok_g = ctx.find_line_segment_groups(img, ml).tobytes() == O.find_line_segment_groups(img, ml, seed=0, num_threads=16)[0].tobytes()
    if not (ok_l and ok_r and ok_g):
        bad += 1
    print("%2d %-12s %4dx%-4d labels %s records %s groups %s  seeds %6d rounds %3d tier2 %5d giants %3d quiet %d" % (
        i, kind, w, h, "ok" if ok_l else "DIFFER", "ok" if ok_r else "DIFFER", "ok" if ok_g else "DIFFER", c["seeds"], c["flood_rounds"],
        c["second_tier_seeds"], c["giant_steps"], c["quiet_round_misses"]), flush=True)
print("%d frames, %d bad, %.0f s" % (n_frames, bad, time.perf_counter() - t_all))
sys.exit(1 if bad else 0)
