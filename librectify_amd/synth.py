"""Seeded synthetic frames for tests and bench.py (SURVEY.md §8d "Synthetic inputs").

frame(W, H, seed): float32 in ~[0,1]: background 0.5 plus K straight bars drawn from three
pencils (two finite vanishing points outside the frame and one near-vertical), edge contrast
U(0.1, 0.5), blurred by sigma = 1, plus N(0, 0.005) noise so that magnitude ties have measure
zero.  Pure numpy; deterministic for a given (W, H, seed, K).
"""
import numpy as np


def _gauss_blur(img, sigma=1.0):
    """Separable blur with edge padding.  Large frames are processed in row chunks on a few threads (numpy
    releases the GIL); every output element is the same sum in the same order either way."""
    r = int(3 * sigma + 0.5)
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-(x * x) / (2 * sigma * sigma))
    k /= k.sum()
    H, W = img.shape
    tmp = np.empty_like(img)
    out = np.empty_like(img)

    def horiz(a, b):
        pad = np.pad(img[a:b], ((0, 0), (r, r)), mode="edge")
        acc = np.zeros((b - a, W), img.dtype)
        for i, kv in enumerate(k):
            acc += kv * pad[:, i : i + W]
        tmp[a:b] = acc

    def vert(a, b):
        rows = np.clip(np.arange(a - r, b + r), 0, H - 1)  # edge padding in rows
        pad = tmp[rows]
        acc = np.zeros((b - a, W), img.dtype)
        for i, kv in enumerate(k):
            acc += kv * pad[i : i + (b - a), :]
        out[a:b] = acc

    chunks = [(a, min(H, a + 256)) for a in range(0, H, 256)]
    if len(chunks) < 4:
        for f in (horiz, vert):
            for a, b in chunks:
                f(a, b)
        return out
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(8) as ex:
        for f in (horiz, vert):
            list(ex.map(lambda ab: f(*ab), chunks))
    return out


def default_bars(W, H):
    return max(12, int(round(320 * (W * H / (3840.0 * 2160.0)) ** 0.5)))


def frame(W, H, seed, bars=None, noise=0.005, tile=None):
    """tile: if given (e.g. 512), bars are confined to tile x tile blocks ("aerial style")."""
    rng = np.random.RandomState(seed)
    K = default_bars(W, H) if bars is None else bars
    img = np.full((H, W), 0.5, np.float64)
    diag = float(np.hypot(W, H))
    vps = [
        np.array([W / 2 + rng.uniform(1.2, 2.5) * diag, H / 2 + rng.uniform(-0.3, 0.3) * diag]),
        np.array([W / 2 - rng.uniform(1.2, 2.5) * diag, H / 2 + rng.uniform(-0.3, 0.3) * diag]),
        np.array([W / 2 + rng.uniform(-0.2, 0.2) * diag, H / 2 - rng.uniform(3.0, 6.0) * diag]),
    ]
    for _ in range(K):
        vp = vps[rng.randint(0, 3)]
        if tile:
            tx = rng.randint(0, max(1, W // tile)) * tile
            ty = rng.randint(0, max(1, H // tile)) * tile
            c = np.array([tx + rng.uniform(0.1, 0.9) * min(tile, W - tx), ty + rng.uniform(0.1, 0.9) * min(tile, H - ty)])
            length = rng.uniform(0.15, 0.6) * tile
        else:
            c = np.array([rng.uniform(0.05, 0.95) * W, rng.uniform(0.05, 0.95) * H])
            length = rng.uniform(0.03, 0.22) * max(W, H)
        d = vp - c
        d /= np.linalg.norm(d)
        nrm = np.array([-d[1], d[0]])
        half_w = rng.uniform(2.0, 0.004 * max(W, H) + 4.0)
        contrast = rng.uniform(0.1, 0.5) * (1 if rng.rand() < 0.5 else -1)
        ext = length / 2 + half_w + 2
        x0, x1 = int(max(0, c[0] - ext)), int(min(W, c[0] + ext + 1))
        y0, y1 = int(max(0, c[1] - ext)), int(min(H, c[1] + ext + 1))
        if x1 <= x0 or y1 <= y0:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        px, py = xx - c[0], yy - c[1]
        along = px * d[0] + py * d[1]
        across = px * nrm[0] + py * nrm[1]
        m = (np.abs(along) <= length / 2) & (np.abs(across) <= half_w)
        img[y0:y1, x0:x1][m] += contrast
    img = np.clip(img, 0.0, 1.0)
    img = _gauss_blur(img, 1.0)
    if noise > 0:
        img = img + rng.normal(0.0, noise, size=img.shape)
    return img.astype(np.float32)


def random_segments(n, seed, frac_on_pencils=0.6, size=1000.0):
    """Synthetic LineSegment rows for the RANSAC micro-benchmark: 60 % on 3 pencils, 40 % uniform."""
    rng = np.random.RandomState(seed)
    from . import LINE_DTYPE

    out = np.zeros(n, LINE_DTYPE)
    vps = [np.array([3.1 * size, 0.4 * size]), np.array([-2.2 * size, 0.7 * size]), np.array([0.45 * size, -5.0 * size])]
    for i in range(n):
        c = rng.uniform(0.05, 0.95, 2) * size
        L = rng.uniform(0.02, 0.15) * size
        if rng.rand() < frac_on_pencils:
            d = vps[rng.randint(0, 3)] - c
            d /= np.linalg.norm(d)
            ang = rng.normal(0, 0.004)
            d = np.array([d[0] * np.cos(ang) - d[1] * np.sin(ang), d[0] * np.sin(ang) + d[1] * np.cos(ang)])
        else:
            a = rng.uniform(0, np.pi)
            d = np.array([np.cos(a), np.sin(a)])
        p1, p2 = c - d * L / 2, c + d * L / 2
        out[i] = (p1[0], p1[1], p2[0], p2[1], rng.uniform(0.05, 0.5), rng.uniform(0.1, 0.8), -1)
    return out


def region_frame(W, H, seed):
    """A frame WITHOUT strong edges (soft blobs on a ramp, blurred): single floods cover smooth regions of hundreds of
    thousands of pixels -- the content the ordered flood likes least (tools/soak_regions.py, bench.py `worst_case`)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.full((H, W), 0.4, np.float64)
    for _ in range(rng.randint(6, 30)):
        cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(30, 0.2 * W)
        img += rng.uniform(0.05, 0.3) * np.exp(-(((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r)))
    img += rng.uniform(0, 0.3) * xx / W + rng.uniform(0, 0.2) * yy / H
    img = _gauss_blur(np.clip(img, 0, 1), rng.uniform(1.0, 3.0)) + rng.normal(0, rng.uniform(0.001, 0.005), size=img.shape)
    return img.astype(np.float32)


def ramp_frame(W, H, seed=77):
    """No edges at all: a smooth ramp with a slow wave under blurred noise.  Every weak seed reaches regions of 100 000 pixels
    and more, and what each finally gets is a few dozen pixels: before walks beyond the second tier's table were held back
    (kernels_flood.hip: kCtrlLowest) this 4K frame took 279 ms (tools/run_edgeless.py, bench.py `worst_case`)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    ramp = 0.3 + 0.3 * xx / W + 0.1 * np.sin(yy / 300.0) + rng.normal(0, 0.002, size=(H, W))
    return _gauss_blur(ramp, 2.0).astype(np.float32)


def long_bar_frame(W, H, seed, K=60):
    """Bars that run across most of the frame (edges of 2000-3600 px at 4K): walks of several hundred tiles each
    (tools/run_long.py, bench.py `worst_case`)."""
    rng = np.random.RandomState(seed)
    img = np.full((H, W), 0.5, np.float64)
    for _ in range(K):
        c = np.array([rng.uniform(0.3, 0.7) * W, rng.uniform(0.1, 0.9) * H])
        ang = rng.uniform(-0.25, 0.25) + (np.pi / 2 if rng.rand() < 0.3 else 0.0)
        d = np.array([np.cos(ang), np.sin(ang)])
        nrm = np.array([-d[1], d[0]])
        length = rng.uniform(0.5, 0.95) * (W if abs(d[0]) > 0.7 else H)
        half_w = rng.uniform(3.0, 12.0)
        contrast = rng.uniform(0.1, 0.4) * (1 if rng.rand() < 0.5 else -1)
        ext_x = abs(d[0]) * length / 2 + abs(nrm[0]) * half_w + 2
        ext_y = abs(d[1]) * length / 2 + abs(nrm[1]) * half_w + 2
        x0, x1 = int(max(0, c[0] - ext_x)), int(min(W, c[0] + ext_x + 1))
        y0, y1 = int(max(0, c[1] - ext_y)), int(min(H, c[1] + ext_y + 1))
        if x1 <= x0 or y1 <= y0:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        px, py = xx - c[0], yy - c[1]
        m = (np.abs(px * d[0] + py * d[1]) <= length / 2) & (np.abs(px * nrm[0] + py * nrm[1]) <= half_w)
        img[y0:y1, x0:x1][m] += contrast
    img = _gauss_blur(np.clip(img, 0, 1), 1.0) + rng.normal(0, 0.005, size=img.shape)
    return img.astype(np.float32)
