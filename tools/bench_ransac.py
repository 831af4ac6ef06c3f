"""RANSAC / PROSAC scoring micro-benchmark: hypotheses per second at N lines (BASELINE metric, second half)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import librectify_amd as L
from librectify_amd import synth


def run(ctx, n, n_iter, reps=5):
    segs = synth.random_segments(n, 42)
    lines = np.ascontiguousarray(segs, L.LINE_DTYPE)
    # bbox normalisation on the host (geometry.cpp:258-282), as estimate_line_pencils does
    xs = np.concatenate([lines["x1"], lines["x2"]]); ys = np.concatenate([lines["y1"], lines["y2"]])
    cx, cy = xs.min() + 0.5 * (xs.max() - xs.min()), ys.min() + 0.5 * (ys.max() - ys.min())
    sc = max(xs.max() - xs.min(), ys.max() - ys.min())
    norm = lines.copy()
    for a, c0 in (("x1", cx), ("x2", cx), ("y1", cy), ("y2", cy)):
        norm[a] = (lines[a] - np.float32(c0)) / np.float32(sc)
    idx = np.arange(n, dtype=np.int32)
    tol = float(np.float32(1.0) - np.float32(np.cos(np.float32(2.0) / np.float32(180.0) * np.float32(np.pi))))
    ctx.ransac_best(norm, idx, tol, n_iter, 42)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.ransac_best(norm, idx, tol, n_iter, 42)
    dt = (time.perf_counter() - t0) / reps
    return n_iter / dt, dt


if __name__ == "__main__":
    ctx = L.Context(0)
    for n, it in [(1000, 10000), (1000, 100000), (20000, 10000), (20000, 100000)]:
        hps, dt = run(ctx, n, it)
        print("N=%d lines, %d hypotheses: %.3f ms per solve (incl. upload + argmax + sync) -> %.3e hypotheses/s, %.3e line evaluations/s" % (n, it, dt * 1e3, hps, hps * n))
