"""Pins the CPU oracle to the reference's own artefacts.  CPU only."""
import os

import numpy as np
import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden_lines():
    rows = np.loadtxt(os.path.join(G, "doc_warp_lines.csv"), delimiter=",")
    assert rows.shape == (848, 7)
    return rows


def test_pin1_transform_kat():
    """doc/image.jpg_warp_lines.csv -> doc/image.jpg_warp_tform.csv, cfg of autorectify.cpp:343-348
    (h_strategy given on the command line as ROTATE_V for the doc artefacts).  6 printed digits."""
    rows = _golden_lines()
    lines = O.lines_from_rows(rows)
    cfg = O.RectificationConfig(40.0, 1.5, O.RECTIFY, 2.0, O.ROTATE_V)
    T = O.compute_rectification_transform(lines, 1000, 563, cfg)
    got = O.transform_to_array(T)
    exp = [l.strip().split(",") for l in open(os.path.join(G, "doc_warp_tform.csv"))]
    exp = [[float(x) for x in r] for r in exp]
    for k in range(4):  # TL, TR, BL, BR
        np.testing.assert_allclose(got[k, :2], exp[k], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(got[4], exp[4], rtol=2e-5, atol=2e-4)  # hvp (ideal)
    np.testing.assert_allclose(got[5], exp[5], rtol=2e-5, atol=2e-4)  # vvp


def test_pin1_group_vps():
    rows = _golden_lines()
    ids, vps = O.fit_vanishing_points(O.lines_from_rows(rows))
    assert list(ids) == [0, 1, 2, 3]
    exp = {0: (445.749, -2111.33), 1: (1620.86, 536.44), 2: (-136.68, 576.30), 3: (251.54, 347.33)}
    for g, v in zip(ids, vps):
        assert v[2] == 1.0
        np.testing.assert_allclose(v[:2], exp[int(g)], rtol=2e-5, atol=0.02)


def test_pin2_detector_structural_kat():
    """doc/image.jpg at TRACE_TOLERANCE=0.3 reproduces >=700 of the 848 golden rows within 0.01 px
    (the doc artefacts predate today's 0.25)."""
    gray = np.load(os.path.join(G, "doc_image_gray.npy"))
    img = gray.astype(np.float32) / np.float32(256.0)
    r = O.find_line_segments(img, tolerance=0.3, want_label=False)
    lines = O.filter_lines(r["lines"], 10.0)
    gold = _golden_lines()[:, :4]
    mine = np.stack([lines["x1"], lines["y1"], lines["x2"], lines["y2"]], 1).astype(np.float64)
    hits = 0
    for g in gold:
        d = np.abs(mine - g).max(axis=1)
        if d.min() <= 0.01:
            hits += 1
    assert hits >= 700, hits


def _golden_hits(lines, gold):
    """which golden rows are reproduced within 0.01 px on all four endpoint coordinates (either endpoint order)"""
    mine = np.stack([lines["x1"], lines["y1"], lines["x2"], lines["y2"]], 1).astype(np.float64)
    swapped = mine[:, [2, 3, 0, 1]]
    hit = np.zeros(len(gold), bool)
    for i, g in enumerate(gold):
        hit[i] = min(np.abs(mine - g).max(axis=1).min(), np.abs(swapped - g).max(axis=1).min()) <= 0.01
    return hit


def test_pin5_refine_structural_kat():
    """Soft pin of `refine` (postprocess_lines_segments, line_detector.cpp:332-444) to the reference's own rows.
    135 of the 848 golden rows are not detector output but MERGED segments, made by the reference's refine with older
    constants than today's.  tools/sweep_refine_pins.py swept the four constants of the pair test
    (today: |cos| >= 0.99, normal offset < 0.02, overlap window (-0.5, 1.5)): with 0.98 / 0.05 / (-0.2, 1.2) the oracle's
    detector (TRACE_TOLERANCE 0.3, as pin 2) -> refine -> filter_lines(10) reproduces 123 of those 135 rows and 791 of
    all 848 within 0.01 px; with today's constants still 49 of the 135.  This pins the pair geometry, the forward-only
    graph walk (:277-329) and merge_lines (:254-274) -- the whole of refine but its four constants."""
    gray = np.load(os.path.join(G, "doc_image_gray.npy"))
    img = gray.astype(np.float32) / np.float32(256.0)
    gold = _golden_lines()[:, :4]
    raw = O.find_line_segments(img, tolerance=0.3, want_label=False)["lines"]
    base = _golden_hits(O.filter_lines(raw, 10.0), gold)
    assert base.sum() >= 700 and (~base).sum() <= 148
    old = _golden_hits(O.filter_lines(O.refine_lines_params(raw, 0.98, 0.05, -0.2, 1.2), 10.0), gold)
    assert (old & ~base).sum() >= 115, (old & ~base).sum()
    assert old.sum() >= 780, old.sum()
    # Round 4: 43 of the 57 rows still unexplained were detector rows that this refine merged with a neighbouring fragment of
    # 3-5 px and the reference's did not.  With fragments of at most 5.9 px kept out of the pair graph (a pre-merge length
    # gate: today's reference has none; LINE_MIN_LENGTH is 5) the same constants reproduce 831 of the 848 rows, and 835 when
    # the lines arrive in a slightly different order (the golden run was threaded: tools/sweep_refine_pins.py order).  The
    # 13 rows left are listed in DESIGN.md section 2.
    gated = _golden_hits(O.filter_lines(O.refine_lines_params(raw, 0.98, 0.05, -0.2, 1.2, 5.9), 10.0), gold)
    assert gated.sum() >= 825, gated.sum()
    assert (gated & ~base).sum() >= 120 and (base & ~gated).sum() <= 10, ((gated & ~base).sum(), (base & ~gated).sum())
    today = _golden_hits(O.filter_lines(O.refine_lines(raw), 10.0), gold)
    assert (today & ~base).sum() >= 45, (today & ~base).sum()
    # the parameterised entry with today's constants is the production refine
    a = O.refine_lines(raw)
    b = O.refine_lines_params(raw)
    assert a.tobytes() == b.tobytes()


def _grouping_agreement(gold_ids, new_ids):
    """per golden group: (label most of its lines got, how many got it, size)"""
    import collections

    out = {}
    for g in (0, 1, 2):
        idx = np.nonzero(gold_ids == g)[0]
        lab, cnt = collections.Counter(new_ids[idx].tolist()).most_common(1)[0]
        out[g] = (lab, cnt, len(idx))
    return out


def test_pin4_grouping_structural_kat():
    """The 848 golden rows carry the group ids the reference's RANSAC peeling gave them (380 / 192 / 80 / 5 lines,
    191 ungrouped).  Grouping the same lines again (estimate_line_pencils, line_pencil.cpp:148-177; the reference seeds
    from random_device, so only the structure can be pinned) must find the same three pencils in the same order:
    >= 94 % of each golden group under one label (363 of 380, 182 of 192, 80 of 80), label k for group k, and vanishing points within 1 % of the golden
    groups' (group 2 coincides exactly: 80 of 80 lines).  Any RANSAC seed."""
    rows = _golden_lines()
    lines = O.lines_from_rows(rows)
    gold = lines["group_id"].copy()
    _, gold_vps = O.fit_vanishing_points(lines)
    for seed in (0, 1, 7):
        blank = lines.copy()
        blank["group_id"] = -1
        got, _ = O.estimate_line_pencils(blank, seed=seed)
        agree = _grouping_agreement(gold, got["group_id"])
        for g, (lab, cnt, size) in agree.items():
            assert lab == g and cnt >= 0.94 * size, (seed, agree)
        _, vps = O.fit_vanishing_points(got)
        for g in (0, 1, 2):
            np.testing.assert_allclose(vps[g][:2], gold_vps[g][:2], rtol=0.01, atol=2.0)


def test_pin3_analytic_kats():
    """src/test.cpp:19-39 inputs; answers derived in SURVEY.md §4."""
    rows = np.array(
        [[0, 0, 10, 0, 1, 0, 10], [10, 0, 8, 5, 1, 0, 1], [8, 5, 2, 5, 1, 0, 10], [2, 5, 0, 0, 1, 0, 1]], np.float64
    )
    ls = O.lines_from_rows(rows)
    vp1 = O.fit_vanishing_point(ls, 1)
    np.testing.assert_allclose(vp1, [5.0, 12.5, 1.0], rtol=1e-4)
    vp2 = O.fit_vanishing_point(ls, 10)
    assert vp2[2] == 0.0
    np.testing.assert_allclose(np.abs(vp2[:2]) / np.linalg.norm(vp2[:2]), [1.0, 0.0], atol=1e-4)
    vp_all = O.fit_vanishing_point(ls, -1)
    vp_zero = O.fit_vanishing_point(ls, 0)  # transform.cpp:35 tests g > 0: group 0 also means "all lines"
    np.testing.assert_array_equal(vp_all, vp_zero)
    probe = O.lines_from_rows(np.array([[5, 1, 5, 4, 1, 0, -1]], np.float64))
    out = O.assign_to_group(ls, probe, 10.0)
    assert out["group_id"][0] == 1


def test_gauss_kernel_matches_formula():
    for dir_x in (True, False):
        K = O.gauss_deriv_kernel(2, 1.0, dir_x)
        j, i = np.meshgrid(np.arange(5) - 2, np.arange(5) - 2)
        z = j if dir_x else i
        ref = z / (2 * np.pi) * np.exp(-(j**2 + i**2) / 2.0)
        np.testing.assert_allclose(K, ref, rtol=1e-6, atol=1e-9)


def test_sampler_distribution_matches_knuth():
    """The counter-based sampler and choice_knuth (math_utils.cpp:14-39) both draw uniform sorted pairs."""
    N = 12
    nd = 60000
    k = O.choice_knuth_mt(123, N, 2, nd)
    assert (k[:, 0] < k[:, 1]).all()
    hk = np.zeros((N, N))
    np.add.at(hk, (k[:, 0], k[:, 1]), 1)
    hc = np.zeros((N, N))
    for it in range(nd):
        a, b = O.sample_pair(7, 0, it, N)
        assert a < b < N
        hc[a, b] += 1
    exp = nd / (N * (N - 1) / 2)
    iu = np.triu_indices(N, 1)
    for hist in (hk, hc):
        chi2 = ((hist[iu] - exp) ** 2 / exp).sum()
        assert chi2 < 120, chi2  # 65 dof; p ~ 1e-5 at 120


def test_oracle_threaded_equals_serial():
    """The oracle's OpenMP regions (placed where the reference has them) write indexed outputs: its results do not
    depend on the thread count, so the large GPU parity tests may run it threaded."""
    from librectify_amd import synth

    img = synth.frame(640, 480, 5)
    a = O.find_line_segments(img, num_threads=-1)
    b = O.find_line_segments(img, num_threads=O.max_threads())
    assert a["lines"].tobytes() == b["lines"].tobytes()
    np.testing.assert_array_equal(a["label"], b["label"])
    fa, _ = O.find_line_segment_groups(img, 6.4, seed=0, num_threads=-1)
    fb, _ = O.find_line_segment_groups(img, 6.4, seed=0, num_threads=O.max_threads())
    assert fa.tobytes() == fb.tobytes()
    for refine in (True,):
        ra, _ = O.find_line_segment_groups(img, 6.4, refine=refine, seed=0, num_threads=-1)
        rb, _ = O.find_line_segment_groups(img, 6.4, refine=refine, seed=0, num_threads=O.max_threads())
        assert ra.tobytes() == rb.tobytes()


def test_cht_estimator_oracle_self_consistency_parity_unpinned():
    """The oracle's diamond-space peeling (what cht.h:13-24 describes inside estimator.h:99-145): finds the three
    pencils of the synthetic segments, is deterministic, and a finer accumulator agrees on the large groups."""
    from librectify_amd import synth

    segs = synth.random_segments(1000, 42)
    a, ma, ca = O.estimate_line_pencils_cht(segs, d=128)
    b, mb, cb = O.estimate_line_pencils_cht(segs, d=128)
    assert a.tobytes() == b.tobytes() and (ca == cb).all()
    ids, cnt = np.unique(a["group_id"], return_counts=True)
    assert set(ids.tolist()) >= {0, 1, 2}
    assert sorted(cnt[ids >= 0])[-3] > 120  # three pencils of ~200 lines each
    assert len(ma) == len(ca) <= 4 and np.allclose(np.linalg.norm(ma, axis=1), 1.0, atol=1e-5)
    # fewer than two lines: no round at all (estimator.h:115)
    one, m1, c1 = O.estimate_line_pencils_cht(segs[:1])
    assert len(m1) == 0 and one["group_id"][0] == -1


def test_separable_gradient_stays_within_ulps_of_the_reference_25_tap_form():
    """the canonical gradient of this build is the separable evaluation of the reference's taps (which the
    HIP kernel mirrors); the reference itself sums 25 separately rounded taps (filter.cpp:65-98).  The two are kept side
    by side in the oracle: on the doc image and on a synthetic frame they differ by a few ulp of the largest magnitude,
    the bin decision agrees wherever the two best directional responses are not within that distance of each other, and
    the seeds are the same but for such ties."""
    from librectify_amd import synth

    gray = np.load(os.path.join(G, "doc_image_gray.npy"))
    for img in (gray.astype(np.float32) / np.float32(256.0), synth.frame(640, 480, 3)):
        f = O.filter_stage(img, planes=True)
        dx25, dy25 = O.conv_gradients_25tap(img)
        scale = float(max(np.abs(f["dx"]).max(), np.abs(f["dy"]).max()))
        ulp = scale * 2.0 ** -23
        assert np.abs(f["dx"] - dx25).max() <= 8 * ulp, np.abs(f["dx"] - dx25).max() / ulp
        assert np.abs(f["dy"] - dy25).max() <= 8 * ulp, np.abs(f["dy"] - dy25).max() / ulp
        assert (dx25[:2] == 0).all() and (dx25[-2:] == 0).all() and (dx25[:, :2] == 0).all() and (dx25[:, -2:] == 0).all()
        # bins from the 25-tap gradients: the same first-strict-argmax rule
        st, ct = O.bin_trig()
        planes25 = np.abs(dx25[None] * st[:, None, None] + dy25[None] * ct[:, None, None])
        bin25 = planes25.argmax(axis=0)
        differ = bin25 != f["bin"]
        # (the masked planes of the oracle are zero outside their dilated masks: the unmasked 25-tap responses give the gap)
        srt = np.sort(planes25, axis=0)
        gap = srt[-1] - srt[-2]
        assert (gap[differ] <= 32 * ulp).all(), float(gap[differ].max() / ulp)
        assert differ.mean() < 2e-3
        mag25 = np.sqrt(dx25 * dx25 + dy25 * dy25)
        s_can = O.find_seeds(f["mag"], f["bin"])
        s_25 = O.find_seeds(mag25.astype(np.float32), bin25.astype(np.int32))
        a = set(zip(s_can["rows"].tolist(), s_can["cols"].tolist()))
        b = set(zip(s_25["rows"].tolist(), s_25["cols"].tolist()))
        assert len(a ^ b) <= 0.02 * len(a), (len(a), len(b), len(a ^ b))
