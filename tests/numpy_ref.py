"""A SECOND, independent restatement of the host rows of the path — fit_optimal (A17), fit_vanishing_points (A23),
select_vertical/horizontal_point (A24), compute_image_transform (A25), compute_rectification_transform (A26) — in
float64 NumPy, written from the reference's text (line_pencil.cpp:111-128, transform.cpp:24-211,
interface.cpp:93-208, geometry.cpp:96-112,214-282) in its own matrix form: cross products, an eigen-solver
(numpy.linalg.eigh), a 3x3 inverse (numpy.linalg.inv).  The oracle (oracle/rectify_oracle.cpp) and the product's
vp_host.cpp share their scalar formulas; this file shares nothing with either, so agreement with it is not the same
text twice.  Test infrastructure only."""
import numpy as np

EPS = 1e-6  # config.h:59
ROTATE_H, ROTATE_V, RECTIFY, KEEP = 0, 1, 2, 3  # librectify.h:126-132


def _endpoints(lines):
    a = np.stack([lines["x1"], lines["y1"]], 1).astype(np.float64)
    b = np.stack([lines["x2"], lines["y2"]], 1).astype(np.float64)
    return a, b


def bbox_normalisation(lines):
    """geometry.cpp:96-112 (bounding_box), :272-282 (centre = min + size / 2, scale = larger side)"""
    a, b = _endpoints(lines)
    pts = np.concatenate([a, b])
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    return lo + 0.5 * (hi - lo), float((hi - lo).max())


def pencil(lines, centre, scale):
    """LinePencilModel (line_pencil.cpp:25-32) of the normalised segments (geometry.cpp:258-270): unit homogeneous lines
    h = (a, 1) x (b, 1) and segment lengths."""
    a, b = _endpoints(lines)
    a = (a - centre) / scale
    b = (b - centre) / scale
    ha = np.concatenate([a, np.ones((len(a), 1))], 1)
    hb = np.concatenate([b, np.ones((len(b), 1))], 1)
    h = np.cross(ha, hb)
    nrm = np.linalg.norm(h, axis=1, keepdims=True)
    h = np.where(nrm > 0, h / np.where(nrm > 0, nrm, 1), h)
    return h, np.linalg.norm(b - a, axis=1)


def fit_optimal(h, length, idx=None):
    """line_pencil.cpp:111-128: cov = h^T diag(length) h over the index set (an empty set means every line); eigenvector
    of the smallest eigenvalue."""
    if idx is not None and len(idx) > 0:
        h, length = h[idx], length[idx]
    cov = h.T @ (h * length[:, None])
    w, v = np.linalg.eigh(cov)
    return v[:, int(np.argmin(w))]


def normalize_point(p):
    """geometry.cpp:232-238"""
    if abs(p[2]) < EPS:
        return np.array([p[0], p[1], 0.0])
    return np.array([p[0] / p[2], p[1] / p[2], 1.0])


def fit_vanishing_points(lines):
    """transform.cpp:52-81: one refit per distinct group id other than -1, in ascending id order (std::map); finite points
    back to image coordinates."""
    centre, scale = bbox_normalisation(lines)
    h, length = pencil(lines, centre, scale)
    out = {}
    for g in sorted(set(int(x) for x in lines["group_id"]) - {-1}):
        vp = normalize_point(fit_optimal(h, length, np.nonzero(lines["group_id"] == g)[0]))
        if vp[2] > 0:
            vp[:2] = scale * vp[:2] + centre
        out[g] = vp
    return out


def fit_single_vanishing_point(lines, g):
    """transform.cpp:24-47 (note `g > 0`: groups 0 and -1 both mean every line)"""
    centre, scale = bbox_normalisation(lines)
    h, length = pencil(lines, centre, scale)
    idx = np.nonzero(lines["group_id"] == g)[0] if g > 0 else None
    vp = normalize_point(fit_optimal(h, length, idx))
    if vp[2] > 0:
        vp[:2] = scale * vp[:2] + centre
    return vp


def direction(a, b):
    """geometry.cpp:240-245"""
    v = np.array([a[0] - b[0] * a[2], a[1] - b[1] * a[2]])
    n = np.linalg.norm(v)
    return v / n if n > 0 else v


def distance(a, b):
    """geometry.cpp:247-256"""
    if a[2] < EPS or b[2] < EPS:
        return np.inf
    return float(np.hypot(a[0] - b[0], a[1] - b[1]))


def select_vertical_point(vps, centre, angular_tolerance, min_distance, margins=None):
    """transform.cpp:136-170: first point within the angular tolerance of the vertical AND farther than min_distance"""
    thr = np.cos(angular_tolerance / 180.0 * np.pi)
    for v in vps:
        score = abs(direction(v, centre) @ np.array([0.0, 1.0]))
        dist = distance(v, centre)
        if margins is not None:
            margins.append(abs(score - thr))
            if np.isfinite(dist):
                margins.append(abs(dist - min_distance) / max(min_distance, 1.0))
        if score > thr and dist > min_distance:
            return np.array(v, np.float64)
    return np.array([0.0, 1.0, 0.0])


def select_horizontal_point(vps, centre, vertical, min_distance, margins=None):
    """transform.cpp:173-211: first other point with -0.7 < cos(angle to the vertical direction) < 0.05, far enough"""
    vd = direction(vertical, centre)
    for v in vps:
        if np.array_equal(np.asarray(v, np.float64), np.asarray(vertical, np.float64)):
            continue
        score = float(direction(v, centre) @ vd)
        dist = distance(v, centre)
        if margins is not None:
            margins += [abs(score - 0.05), abs(score + 0.7)]
            if np.isfinite(dist):
                margins.append(abs(dist - min_distance) / max(min_distance, 1.0))
        if -0.7 < score < 0.05 and dist > min_distance:
            return np.array(v, np.float64)
    return np.array([1.0, 0.0, 0.0])


def compute_image_transform(width, height, vp_h, vp_v):
    """transform.cpp:84-133: H = [I; l_inf / l_inf.z], affine from the post-H vanishing directions, four centred corners
    (0,0) (W,0) (W,H) (0,H) warped, divided, un-centred.  Rows: TL, TR, BR, BL."""
    vl = np.cross(vp_h, vp_v)
    H = np.eye(3)
    H[2] = vl / vl[2]
    hp, vpp = H @ vp_h, H @ vp_v
    if hp[0] < 0:
        hp = -hp
    if vpp[1] < 0:
        vpp = -vpp
    A1 = np.eye(3)
    A1[:2, 0] = hp[:2] / np.linalg.norm(hp[:2])
    A1[:2, 1] = vpp[:2] / np.linalg.norm(vpp[:2])
    M = np.linalg.inv(A1) @ H
    coords = np.array([[0, width, width, 0], [0, 0, height, height], [1, 1, 1, 1]], np.float64)
    coords[0] -= width / 2.0
    coords[1] -= height / 2.0
    wc = M @ coords
    wc = wc / wc[2]
    wc[0] += width / 2.0
    wc[1] += height / 2.0
    return wc.T


def compute_rectification_transform(lines, width, height, cfg, margins=None):
    """interface.cpp:122-208.  cfg = (vertical_vp_angular_tolerance, vertical_vp_min_distance, v_strategy,
    horizontal_vp_min_distance, h_strategy).  Returns rows TL, TR, BL, BR, hvp, vvp (the order of the doc tform csv)."""
    tol, vmin, vs, hmin, hs = cfg
    vps = list(fit_vanishing_points(lines).values())
    centre = np.array([width / 2.0, height / 2.0, 1.0])
    diag = float(np.linalg.norm(centre[:2]))
    vp_v = select_vertical_point(vps, centre, tol, max(vmin, 1.0) * diag, margins)
    vp_h = select_horizontal_point(vps, centre, vp_v, max(hmin, 1.0) * diag, margins)
    v1, v2 = vp_h.copy(), vp_v.copy()
    if v1[2] != 0:
        v1[:2] -= centre[:2]
    if v2[2] != 0:
        v2[:2] -= centre[:2]
    v1h = v1.copy()
    if hs == ROTATE_H:
        v1h[2] = 0
    elif hs == ROTATE_V:
        v1h = np.array([-v2[1], v2[0], 0.0])
    elif hs != RECTIFY:
        v1h = np.array([1.0, 0.0, 0.0])
    v2h = v2.copy()
    if vs == ROTATE_H:
        v2h = np.array([-v1[1], v1[0], 0.0])
    elif vs == ROTATE_V:
        v2h[2] = 0
    elif vs != RECTIFY:
        v2h = np.array([0.0, 1.0, 0.0])
    t = compute_image_transform(width, height, v1h, v2h)
    if v1h[2] != 0:
        v1h[:2] += centre[:2]
    if v2h[2] != 0:
        v2h[:2] += centre[:2]
    return np.stack([t[0], t[1], t[3], t[2], v1h, v2h])


# ---- refine: postprocess_lines_segments (line_detector.cpp:253-444), second source ----------------------------------
# Written from the reference's text in matrix form, float64.  Returns the merged segments as rows
# (x1, y1, x2, y2, weight, err) and, for the test, how close the nearest pair came to one of the three gates.
def _fit_line_parameters(X, w):
    """geometry.cpp:20-61: weighted principal axis of the points X (rows (row, col)), end points = extreme projections"""
    wn = w / w.sum()
    a = (X * wn[:, None]).sum(axis=0)
    Cn = X - a
    cov = Cn.T @ (wn[:, None] * Cn)
    ev, evec = np.linalg.eigh(cov)  # ascending: column 1 is the major axis, column 0 the normal
    d, n = evec[:, 1], evec[:, 0]
    t = Cn @ d
    c0, c1 = a + d * t.min(), a + d * t.max()
    return np.array([c0[1], c0[0], c1[1], c1[0], w.mean(), np.abs(Cn @ n).mean()])


def refine(lines, cos_gate=0.99, max_offset=0.02, lo=-0.5, hi=1.5):
    P1, P2 = _endpoints(lines)
    n_lines = len(P1)
    dv = P2 - P1
    ln = np.linalg.norm(dv, axis=1)
    d = dv / ln[:, None]
    nrm = np.stack([-d[:, 1], d[:, 0]], axis=1)
    weight = np.asarray(lines["weight"], np.float64)
    closest = np.inf  # smallest distance of a DECISIVE test quantity from its gate: one whose other two gates are open
    succ = [[] for _ in range(n_lines)]
    for i in range(n_lines - 1):
        js = np.arange(i + 1, n_lines)
        cosv = np.abs(d[js] @ d[i])
        for j, cv in zip(js[cosv >= cos_gate - 1e-3], cosv[cosv >= cos_gate - 1e-3]):
            if ln[i] < ln[j]:  # the shorter one's end points in the longer one's frame, in units of its length
                Wm = (np.stack([P1[i], P2[i]]) - P1[j]) @ np.stack([d[j], nrm[j]], axis=1) / ln[j]
            else:
                Wm = (np.stack([P1[j], P2[j]]) - P1[i]) @ np.stack([d[i], nrm[i]], axis=1) / ln[i]
            off = np.abs(Wm[:, 1]).max()
            x = Wm[:, 0]
            g_cos, g_off, g_ovl = cv >= cos_gate, off < max_offset, bool((x > lo).any() and (x < hi).any())
            if g_off and g_ovl:
                closest = min(closest, abs(cv - cos_gate))
            if g_cos and g_ovl:
                closest = min(closest, abs(off - max_offset))
            if g_cos and g_off:
                # (the overlap gate is "any end point above lo and any below hi": it flips where the larger crosses lo or
                # the smaller crosses hi)
                closest = min(closest, abs(x.max() - lo), abs(x.min() - hi))
            if g_cos and g_off and g_ovl:
                succ[i].append(int(j))
    # graph_components / dfs (line_detector.cpp:282-329): a breadth-first walk that only follows edges to HIGHER indices
    comp = -np.ones(n_lines, np.int64)
    visited = np.zeros(n_lines, bool)
    for v in range(n_lines):
        if visited[v]:
            continue
        queue = [v]
        while queue:
            u = queue.pop(0)
            visited[u] = True
            comp[u] = v
            queue.extend(j for j in succ[u] if not visited[j])
    out = []
    for lbl in sorted(set(comp.tolist())):
        idx = np.nonzero(comp == lbl)[0]
        if len(idx) == 1:
            k = idx[0]
            out.append([P1[k][0], P1[k][1], P2[k][0], P2[k][1], weight[k], float(lines["err"][k])])
            continue
        wts = ln[idx] * weight[idx]  # merge_lines (line_detector.cpp:253-274)
        X = np.empty((2 * len(idx), 2))
        X[0::2] = P1[idx][:, ::-1]
        X[1::2] = P2[idx][:, ::-1]
        m = _fit_line_parameters(X, np.repeat(wts, 2))
        m[4] = wts.sum() / ln[idx].sum()
        out.append(m.tolist())
    return np.array(out), float(closest)
