"""Dumps what the flood needs (dx, dy, dmask, seed list) of a synthetic frame to a directory, for tools/sim/flood_sim.
Test/experiment tooling: uses the CPU oracle, never part of the product path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from librectify_amd import synth  # noqa: E402


def main():
    w, h, seed, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    kind = sys.argv[5] if len(sys.argv) > 5 else "synth"
    os.makedirs(out, exist_ok=True)
    if kind == "synth":
        img = synth.frame(w, h, seed)
    elif kind == "tiled":
        img = synth.frame(w, h, seed, bars=int(sys.argv[6]), tile=512)
    elif kind == "long":  # the long-edge stress frame of tools/run_long.py
        rng = np.random.RandomState(seed)
        img = np.full((h, w), 0.5, np.float64)
        yy, xx = np.mgrid[0:h, 0:w]
        for _ in range(60):
            c = np.array([rng.uniform(0.3, 0.7) * w, rng.uniform(0.1, 0.9) * h])
            ang = rng.uniform(-0.25, 0.25) + (np.pi / 2 if rng.rand() < 0.3 else 0.0)
            d = np.array([np.cos(ang), np.sin(ang)]); nrm = np.array([-d[1], d[0]])
            length = rng.uniform(0.5, 0.95) * (w if abs(d[0]) > 0.7 else h)
            half_w = rng.uniform(3.0, 12.0)
            contrast = rng.uniform(0.1, 0.4) * (1 if rng.rand() < 0.5 else -1)
            px, py = xx - c[0], yy - c[1]
            m = (np.abs(px * d[0] + py * d[1]) <= length / 2) & (np.abs(px * nrm[0] + py * nrm[1]) <= half_w)
            img[m] += contrast
        img = (synth._gauss_blur(np.clip(img, 0, 1), 1.0) + rng.normal(0, 0.005, size=img.shape)).astype(np.float32)
    elif kind == "doc":  # the natural 4K frame of tools/run_doc4k.py
        import scipy.ndimage as ndi
        g = np.load(os.path.join(ROOT, "tests", "golden", "doc_image_gray.npy")).astype(np.float32) / np.float32(256.0)
        img = np.ascontiguousarray(ndi.zoom(g, (h / g.shape[0], w / g.shape[1]), order=3).astype(np.float32)[:h, :w])
    else:
        raise SystemExit("kind?")
    f = O.filter_stage(img, num_threads=8)
    s = O.find_seeds(f["mag"], f["bin"])
    st, ct = O.bin_trig()
    idx = (s["rows"].astype(np.int64) * w + s["cols"]).astype(np.int32)
    b = s["bins"].astype(np.int32)
    dxs, dys = f["dx"].reshape(-1)[idx], f["dy"].reshape(-1)[idx]
    resp = np.abs(np.float32(dxs * st[b]) + np.float32(dys * ct[b])).astype(np.float32)  # (sim only: not the fused form)
    thr = (np.float32(0.75) * resp).astype(np.float32)
    f["dx"].tofile(os.path.join(out, "dx.f32"))
    f["dy"].tofile(os.path.join(out, "dy.f32"))
    f["dmask"].tofile(os.path.join(out, "dmask.u8"))
    idx.tofile(os.path.join(out, "seed_idx.i32"))
    b.tofile(os.path.join(out, "seed_bin.i32"))
    thr.tofile(os.path.join(out, "seed_thr.f32"))
    np.concatenate([st, ct]).astype(np.float32).tofile(os.path.join(out, "trig.f32"))
    open(os.path.join(out, "meta.txt"), "w").write("%d %d %d\n" % (w, h, len(idx)))
    print(w, h, len(idx))


main()
