"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors.  Bars: bit-exact for dx/dy, masks, seeds, labels, segment records and group ids;
1e-4 (relative, on homogeneous coordinates scaled to unit norm) for vanishing points."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def L():
    import librectify_amd as L

    L.lib()
    assert L.device_count() > 0, "GPU tests need a GPU"
    return L


@pytest.fixture(scope="module")
def ctx(L):
    c = L.Context(0)
    yield c
    c.close()


def _frames():
    from librectify_amd import synth

    return {
        "96x64": synth.frame(96, 64, 11, bars=10),
        "257x131": synth.frame(257, 131, 12, bars=14),  # width not a multiple of 4
        "320x240": synth.frame(320, 240, 13, bars=24),
        "67x45": synth.frame(67, 45, 14, bars=6),  # ragged, smaller than 2 tiles
        "640x480": synth.frame(640, 480, 5),
        "noiseless": synth.frame(200, 150, 21, bars=12, noise=0.0),  # exact zeros and magnitude ties
    }


FRAMES = _frames()


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _assert_lines_equal(a, b):
    assert len(a) == len(b), (len(a), len(b))
    if len(a):
        av = np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint32).reshape(len(a), 7)
        bv = np.frombuffer(np.ascontiguousarray(b).tobytes(), np.uint32).reshape(len(b), 7)
        bad = np.nonzero((av != bv).any(axis=1))[0]
        assert len(bad) == 0, "first mismatch at %d: %s vs %s" % (bad[0], a[bad[0]], b[bad[0]])


@pytest.mark.parametrize("name", list(FRAMES))
def test_filter_stage_bit_exact(L, ctx, name):
    img = FRAMES[name]
    ref = O.filter_stage(img)
    ctx.stage_filter_host(img)
    dx = ctx.download(L.BUF_DX)
    dy = ctx.download(L.BUF_DY)
    dm = ctx.download(L.BUF_DMASK)
    assert (_bits(dx) != _bits(ref["dx"])).sum() == 0
    assert (_bits(dy) != _bits(ref["dy"])).sum() == 0
    np.testing.assert_array_equal(dm, ref["dmask"])


@pytest.mark.parametrize("name", list(FRAMES))
def test_seeds_match(L, ctx, name):
    img = FRAMES[name]
    h, w = img.shape
    ref = O.filter_stage(img, planes=True)
    seeds = O.find_seeds(ref["mag"], ref["bin"])
    ctx.stage_filter_host(img)
    n = ctx.stage_seeds()
    assert n == len(seeds["rows"])
    assert ctx.download(L.BUF_MAXMAG)[0] == ref["mag"].max()
    idx = ctx.download(L.BUF_SEED_IDX)
    np.testing.assert_array_equal(idx, seeds["rows"] * w + seeds["cols"])
    np.testing.assert_array_equal(ctx.download(L.BUF_SEED_BIN), seeds["bins"])
    thr = np.float32(0.75) * ref["planes"][seeds["bins"], seeds["rows"], seeds["cols"]]
    assert (_bits(ctx.download(L.BUF_SEED_THR)) != _bits(thr.astype(np.float32))).sum() == 0


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("name", list(FRAMES))
def test_flood_labels_and_segments_bit_exact(L, ctx, name, mode):
    img = FRAMES[name]
    ref = O.find_line_segments(img)
    ctx.set_flood_mode(mode)
    ctx.stage_filter_host(img)
    ctx.stage_seeds()
    ctx.stage_flood()
    lab = ctx.download(L.BUF_LABEL)
    bad = np.argwhere(lab != ref["label"])
    assert len(bad) == 0, "%d label mismatches, first at %s: gpu %d oracle %d" % (
        len(bad), bad[0], lab[tuple(bad[0])], ref["label"][tuple(bad[0])])
    lines = ctx.stage_fit()
    _assert_lines_equal(lines, ref["lines"])
    ctx.set_flood_mode(1)


def _unit(v):
    v = np.asarray(v, np.float64)
    return v / np.linalg.norm(v)


@pytest.mark.parametrize("name", ["96x64", "257x131", "320x240", "640x480"])
def test_full_path_matches_oracle(L, ctx, name):
    img = FRAMES[name]
    h, w = img.shape
    ml = float(max(w, h)) / 100.0
    ref, _ = O.find_line_segment_groups(img, ml, seed=0)
    ctx.set_seed(0)
    got = ctx.find_line_segment_groups(img, ml)
    _assert_lines_equal(got, ref)  # endpoints, weight, err and group ids
    Tg = L.compute_rectification_transform(got, w, h).as_array()
    Tr = O.transform_to_array(O.compute_rectification_transform(ref, w, h))
    for k in (4, 5):
        assert np.abs(_unit(Tg[k]) - _unit(Tr[k])).max() < 1e-4
    np.testing.assert_allclose(Tg[:4], Tr[:4], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("case", ["synth_96x64_s11", "synth_257x131_s12", "synth_320x240_s13"])
def test_against_committed_golden(L, ctx, case):
    z = np.load(os.path.join(G, case + ".npz"))
    img = z["image"]
    h, w = img.shape
    ctx.stage_filter_host(img)
    if "dx" in z:
        assert (_bits(ctx.download(L.BUF_DX)) != _bits(z["dx"])).sum() == 0
        assert (_bits(ctx.download(L.BUF_DY)) != _bits(z["dy"])).sum() == 0
    np.testing.assert_array_equal(ctx.download(L.BUF_DMASK), z["dmask"])
    assert ctx.stage_seeds() == len(z["seed_idx"])
    np.testing.assert_array_equal(ctx.download(L.BUF_SEED_IDX), z["seed_idx"])
    ctx.stage_flood()
    np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), z["label"])
    _assert_lines_equal(ctx.stage_fit(), z["raw_lines"].view(L.LINE_DTYPE).reshape(-1))
    ctx.set_seed(0)
    got = ctx.find_line_segment_groups(img, float(max(w, h)) / 100.0)
    _assert_lines_equal(got, z["grouped_lines"].view(L.LINE_DTYPE).reshape(-1))
    T = L.compute_rectification_transform(got, w, h).as_array()
    for k in (4, 5):
        assert np.abs(_unit(T[k]) - _unit(z["transform"][k])).max() < 1e-4


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_doc_image_detector_kat_on_gpu(L, ctx, mode):
    """doc/image.jpg through the HIP detector, compared with the oracle bit for bit (the oracle
    itself is pinned to the reference's golden rows in test_oracle_pins.py)."""
    gray = np.load(os.path.join(G, "doc_image_gray.npy"))
    img = gray.astype(np.float32) / np.float32(256.0)
    ref = O.find_line_segments(img)
    ctx.set_flood_mode(mode)
    ctx.stage_filter_host(img)
    assert ctx.stage_seeds() == ref["n_seeds"] == 6649
    ctx.stage_flood()
    np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
    lines = ctx.stage_fit()
    _assert_lines_equal(lines, ref["lines"])
    assert len(lines) == 1927
    ctx.set_flood_mode(1)


def test_ransac_scoring_matches_oracle(L, ctx):
    from librectify_amd import synth

    for n, n_iter, seed in [(50, 500, 1), (333, 2000, 42), (1000, 10000, 7)]:
        segs = synth.random_segments(n, seed)
        norm, _, _ = O.normalize_lines(segs)
        idx = np.arange(n, dtype=np.int32)[:: 1 if n < 100 else 2]  # a strict subset as in later peeling rounds
        tol = O.cos_threshold(2.0)
        ref = O.ransac_best(norm, idx, tol, n_iter, seed, rnd=1)
        got = ctx.ransac_best(norm, idx, tol, n_iter, seed, rnd=1)
        assert got["iter"] == ref["iter"]
        assert np.float32(got["score"]) == np.float32(ref["score"])
        np.testing.assert_array_equal(got["best_h"], ref["best_h"])
        a, _ = O.estimate_line_pencils(segs, n_iter=n_iter, seed=seed)
        b = ctx.estimate_line_pencils(segs, n_iter=n_iter, seed=seed)
        np.testing.assert_array_equal(a["group_id"], b["group_id"])


def test_hough_weights_and_prosac_match_oracle(L, ctx):
    """Opt-in PROSAC path (prosac.h, line_pencil.cpp:47-86): self-golden against the oracle (the reference never
    instantiates it).  Includes BASELINE configs[4]'s shape: ~20k segments, 100k hypotheses."""
    from librectify_amd import synth

    for n, T_N, seed in [(300, -1, 3), (1000, 2000, 42), (20000, 100000, 7)]:
        segs = synth.random_segments(n, seed)
        norm, _, _ = O.normalize_lines(segs)
        idx = np.arange(n, dtype=np.int32)
        tol = O.cos_threshold(2.0)
        wr = O.get_weights_fixed(norm, idx)
        wg = ctx.ht_weights(norm, idx)
        assert (_bits(wg) != _bits(wr)).sum() == 0
        ref = O.prosac_solve(norm, idx, tol, T_N, seed=seed)
        got = ctx.prosac_solve(norm, idx, tol, T_N, seed=seed)
        for k in ("iterations", "n_star", "best_iter", "I_N_best"):
            assert got[k] == ref[k], (k, got[k], ref[k])
        np.testing.assert_array_equal(got["h"], ref["h"])
    segs = synth.random_segments(1500, 5)
    a = O.estimate_line_pencils_prosac(segs, T_N=3000, seed=9)
    b = ctx.estimate_line_pencils_prosac(segs, T_N=3000, seed=9)
    np.testing.assert_array_equal(a["group_id"], b["group_id"])


def test_second_laps_of_the_one_wait_pipeline_are_exact(L, ctx):
    """A frame is one chain of kernels with one host wait; two things make it take a second lap, both detected after
    that wait: more seeds than the seed sort was sized for, and a flood that needs more rounds than were enqueued
    blindly.  Both forced here (and both at once), full path against the oracle; then the next frame runs in one lap.
    (Blind rounds are what the lanes of a batch call enqueue; a single call enqueues its later rounds just in time --
    switched off here, and checked against the same records at the end.)"""
    img = FRAMES["640x480"]
    ref, _ = O.find_line_segment_groups(img, 6.4, seed=0)
    ctx.set_seed(0)
    ctx.set_flood_mode(1)
    ctx.set_flood_just_in_time(False)
    _assert_lines_equal(ctx.find_line_segment_groups(img, 6.4), ref)
    n_seeds = ctx.stage_counters()["seeds"]
    rounds = ctx.stage_counters()["flood_rounds"]
    assert n_seeds > 4096 // 4 and rounds >= 3
    for cap, blind in [(n_seeds // 3, None), (None, 1), (n_seeds // 2, 1)]:
        if cap:
            ctx.set_seed_capacity(cap)
        if blind:
            ctx.set_flood_blind_rounds(blind)
        _assert_lines_equal(ctx.find_line_segment_groups(img, 6.4), ref)
        assert ctx.stage_counters()["frame_laps"] >= 2, (cap, blind, ctx.stage_counters())
        _assert_lines_equal(ctx.find_line_segment_groups(img, 6.4), ref)
        assert ctx.stage_counters()["frame_laps"] == 1
    # the staged API takes the same route when the capacity is too small
    ctx.set_seed_capacity(n_seeds // 3)
    ctx.stage_filter_host(img)
    assert ctx.stage_seeds() == n_seeds
    ctx.stage_flood()
    np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), O.find_line_segments(img)["label"])
    # just in time: however few rounds the first batch has, the frame never takes a second lap for the flood's sake
    ctx.set_flood_just_in_time(True)
    for _ in range(3):
        _assert_lines_equal(ctx.find_line_segment_groups(img, 6.4), ref)
        assert ctx.stage_counters()["frame_laps"] == 1 and ctx.stage_counters()["flood_rounds"] == rounds


def test_seed_order_with_a_capacity_that_cuts_the_last_sort_block(L, ctx):
    """the fused seed order sorts blocks of 4096 keys; a capacity that is no multiple of 4096, with
    the seed count in the last, partial block, used to rank against slots beyond the capacity (stale keys of an earlier
    frame, or memory past the allocation).  A larger frame first leaves real keys beyond the slot, then the frame runs
    with capacity = seeds + 37; and a frame of fewer than 4096 pixels on a fresh context (capacity = its pixel count)."""
    from librectify_amd import synth

    big = synth.frame(1600, 1200, 21)
    img = synth.frame(1120, 840, 22)
    ref = O.find_line_segments(img)
    n = ref["n_seeds"]
    assert 4096 < n < 8192, n  # two sort blocks, the second one partial
    ctx.set_flood_mode(1)
    ctx.set_seed(0)
    for cap in (n + 37, n + 1, n):
        ctx.find_line_segment_groups(big, 16.0)  # leaves its keys in every slot up to its own count (> 8192)
        assert ctx.stage_counters()["seeds"] > 8192
        ctx.set_seed_capacity(cap)
        ctx.stage_filter_host(img)
        assert ctx.stage_seeds() == n
        ctx.stage_flood()
        np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
        _assert_lines_equal(ctx.stage_fit(), ref["lines"])
        ctx.find_line_segment_groups(big, 16.0)
        ctx.set_seed_capacity(cap)
        got = ctx.find_line_segment_groups(img, 11.2)
        assert ctx.stage_counters()["frame_laps"] == 1
        _assert_lines_equal(got, O.find_line_segment_groups(img, 11.2, seed=0)[0])
    small = synth.frame(62, 60, 23, bars=4)
    ref_s, _ = O.find_line_segment_groups(small, 5.0, seed=0)
    c2 = L.Context(0)
    try:
        c2.set_seed(0)
        _assert_lines_equal(c2.find_line_segment_groups(small, 5.0), ref_s)
        c2.stage_filter_host(small)
        c2.stage_seeds()
        c2.stage_flood()
        np.testing.assert_array_equal(c2.download(L.BUF_LABEL), O.find_line_segments(small)["label"])
    finally:
        c2.close()


def test_partial_commits_change_the_work_not_the_labels(L, ctx):
    """The flood's partial commits (a blocked seed commits at once the part of its footprint connected to its seed pixel
    through its own stamps: no lower seed can reach it) against the same flood without them: identical label images and
    records (both equal the oracle's), fewer pixels walked, no more rounds.  Also through the exhausted-storage hooks,
    where seeds that own pixels reach the ordered tail."""
    from librectify_amd import synth

    img = synth.frame(1280, 960, 31)
    ref = O.find_line_segment_groups(img, 12.8, seed=0)[0]
    ref_label = O.find_line_segments(img)["label"]
    ctx.set_seed(0)
    stats = {}
    try:
        for on in (False, True):
            ctx.set_flood_partial_commits(on)
            for mode in (1, 2, 3, 5):
                ctx.set_flood_mode(mode)
                ctx.stage_filter_host(img)
                ctx.stage_seeds()
                ctx.stage_flood()
                np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref_label)
                _assert_lines_equal(ctx.find_line_segment_groups(img, 12.8), ref)
                if mode == 1:
                    stats[on] = ctx.stage_counters()
    finally:
        ctx.set_flood_partial_commits(True)
        ctx.set_flood_mode(1)
    assert stats[True]["walked_px"] < stats[False]["walked_px"]
    assert stats[True]["flood_rounds"] <= stats[False]["flood_rounds"]


def test_rewalks_from_the_logs_change_the_time_not_the_labels(L, ctx):
    """Round 4: a finished walk of sixteen tiles or more leaves its footprint as (tile, pixels) records, and the seed's later
    rounds work the footprint out from them -- components of the records' pixels that are still there, united across tiles in
    LDS (kernels_flood.hip: flood_rewalk_kernel) -- instead of walking again.  Label image and records against the oracle
    with the logs off, on, and on with every log through the fall-back path (sweeps); on frames of bars, long bars (logs of
    the second tier's walks: the larger tables), regions and noise (tiles of many components: the fall-back on its own);
    through the storage hooks.  The counters prove that each path really ran."""
    from librectify_amd import synth

    rng = np.random.RandomState(5)
    noisy = synth.frame(1280, 720, 9, bars=30) + rng.normal(0, 0.02, size=(720, 1280)).astype(np.float32)
    frames = (("bars", synth.frame(1920, 1080, 7)), ("long", synth.long_bar_frame(1920, 1080, 3, K=24)), ("regions", _regions(1280, 720, 4)),
              ("noise", noisy.astype(np.float32)))
    used = {}
    try:
        for name, img in frames:
            ref = O.find_line_segments(img)
            for logs in (1, 0, 2):
                ctx.set_flood_logs(logs)
                for mode in (1, 6, 7) if name in ("bars", "long") else (1,):
                    ctx.set_flood_mode(mode)
                    for rep in range(2):  # (the second frame on the context starts with the hints of the first)
                        ctx.stage_filter_host(img)
                        ctx.stage_seeds()
                        ctx.stage_flood()
                        used[(name, logs, mode)] = ctx.stage_counters()
                        np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
                        _assert_lines_equal(ctx.stage_fit(), ref["lines"])
    finally:
        ctx.set_flood_logs(1)
        ctx.set_flood_mode(1)
    for name, _ in frames:
        assert used[(name, 1, 1)]["log_rewalks"] > 0 and used[(name, 0, 1)]["log_rewalks"] == 0, (name, used[(name, 1, 1)])
        assert used[(name, 2, 1)]["log_give_ups"] == used[(name, 2, 1)]["log_rewalks"] > 0  # (the hook: every log the slow way)
    assert used[("bars", 1, 1)]["log_give_ups"] == 0
    assert used[("long", 1, 1)]["second_tier_seeds"] > 0


def _ramp(W, H, seed):
    from librectify_amd import synth

    return synth.ramp_frame(W, H, seed)


def test_giant_walks_are_held_back_until_they_are_the_lowest(L, ctx):
    """Round 4: a walk that outgrows even the second tier's table no longer moves into a global slab unless it belongs to
    the LOWEST active seed; any other is marked, counts as unfinished, and the window closes behind the lowest marked seed
    plus room for some five hundred marked seeds that walk again each round (kernels_flood.hip: kCtrlLowest).  A valid
    window is any prefix of the seed order, so the labels cannot change: ramp frames against the oracle with the rule
    (mode 1) and without (mode 4: no second tier, slabs at once), twice each on the context (hints of the first frame);
    the counter proves that walks were held, and that mode 1 then needs no slab."""
    used = {}
    try:
        for name, img in (("ramp", _ramp(1920, 1080, 77)), ("ramp2", _ramp(1283, 717, 5)), ("regions", _regions(1920, 1080, 9))):
            ref = O.find_line_segments(img, num_threads=8)
            for mode in (1, 4):
                ctx.set_flood_mode(mode)
                for rep in range(2):
                    ctx.stage_filter_host(img)
                    ctx.stage_seeds()
                    ctx.stage_flood()
                    used[(name, mode)] = ctx.stage_counters()
                    np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
                    _assert_lines_equal(ctx.stage_fit(), ref["lines"])
    finally:
        ctx.set_flood_mode(1)
    assert used[("ramp", 1)]["giants_held"] > 0 and used[("ramp", 1)]["slabs"] <= 1, used[("ramp", 1)]
    assert used[("ramp", 4)]["giants_held"] == 0 and used[("ramp", 4)]["slabs"] > 0, used[("ramp", 4)]
    assert used[("ramp", 1)]["walked_px"] < used[("ramp", 4)]["walked_px"]


def _radial(W, H):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    return (1.0 - np.hypot(xx - W / 2, yy - H / 2) / np.hypot(W / 2, H / 2)).astype(np.float32)


def test_giant_steps_label_what_the_slab_walk_and_the_oracle_label(L, ctx):
    """Round 5: when the lowest active seed's walk outgrows the LDS tiers, its flood -- what the reference's loop does next,
    nothing speculative (line_detector.cpp:98-119 over filter.cpp:110-153) -- is labelled by the whole device between two
    rounds (kernels_flood.hip: kCtrlGiantStep: tile masks, union-find over the tiles' components) instead of being walked
    through a global slab by one team.  Label image and records against the oracle with the step and without (the slab
    walk), rounds enqueued just in time and blindly (the steps then go in from flood_finish's laps), twice each on the
    context; on a noiseless radial gradient (sixteen rings of one magnitude: a chain of sixteen steps, and sixteen
    components of more than 2^14 pixels for the fit's bucket sort), soft blobs, a ramp under noise and diagonal stripes.
    The counters prove which path ran."""
    yy, xx = np.mgrid[0:540, 0:960].astype(np.float64)
    stripes = (0.5 + 0.4 * np.sin((xx + 0.5 * yy) * 2 * np.pi / 40)).astype(np.float32)
    from librectify_amd import synth

    frames = (("radial", _radial(1280, 720)), ("regions", synth.region_frame(1920, 1080, 500)), ("ramp", _ramp(1283, 717, 5)), ("stripes", stripes))
    used = {}
    try:
        for name, img in frames:
            ref = O.find_line_segments(img, num_threads=8)
            for step, jit in ((True, True), (False, True), (True, False)):
                ctx.set_flood_giant_step(step)
                ctx.set_flood_just_in_time(jit)
                for rep in range(2):
                    ctx.stage_filter_host(img)
                    ctx.stage_seeds()
                    ctx.stage_flood()
                    used[(name, step, jit)] = ctx.stage_counters()
                    np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
                    _assert_lines_equal(ctx.stage_fit(), ref["lines"])
            ctx.set_flood_giant_step(True)
            ctx.set_flood_just_in_time(True)
            ctx.set_seed(0)
            _assert_lines_equal(ctx.find_line_segment_groups(img, 12.8), O.find_line_segment_groups(img, 12.8, seed=0, num_threads=8)[0])
    finally:
        ctx.set_flood_giant_step(True)
        ctx.set_flood_just_in_time(True)
    for jit in (True, False):
        assert used[("radial", True, jit)]["giant_steps"] >= 8 and used[("radial", True, jit)]["slabs"] == 0, used[("radial", True, jit)]
    # (blind rounds may stall on the soft blobs -- lists longer than a late round's grid -- and leave them to the ordered tail)
    assert used[("regions", True, True)]["giant_steps"] >= 1 and used[("regions", True, True)]["ordered_tail_seeds"] == 0, used[("regions", True, True)]
    assert used[("radial", False, True)]["giant_steps"] == 0 and used[("radial", False, True)]["slabs"] > 0, used[("radial", False, True)]


def test_rounds_without_the_second_tier_after_a_calm_frame(L, ctx):
    """Round 5: when the context's last frame kept every walk in the first storage tier, the rounds enqueued blindly behind
    the first go without the second tier's launch; a walk that outgrows the first tier there counts as unfinished and the
    rounds enqueued from then on bring the second tier.  A frame of bars (calm) followed by soft blobs, long bars and the
    doc image on the same context: label image and records against the oracle, the counter proves the path ran."""
    from librectify_amd import synth

    g = np.load(os.path.join(ROOT, "tests", "golden", "doc_image_gray.npy")).astype(np.float32) / np.float32(256.0)
    others = (synth.region_frame(1280, 720, 500), synth.long_bar_frame(1280, 720, 5, K=20), np.ascontiguousarray(g))
    misses = 0
    for img in others:
        # (a calm frame of the SAME size: what a context learned goes with the stream, and a frame of another size starts a new one)
        calm = synth.frame(img.shape[1], img.shape[0], 4, bars=40)
        ref_calm = O.find_line_segments(calm)
        ref = O.find_line_segments(img, num_threads=8)
        for frame, r in ((calm, ref_calm), (img, ref), (img, ref)):
            ctx.stage_filter_host(frame)
            ctx.stage_seeds()
            ctx.stage_flood()
            c = ctx.stage_counters()
            np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), r["label"])
            _assert_lines_equal(ctx.stage_fit(), r["lines"])
            if frame is calm:
                assert c["second_tier_seeds"] == 0 and c["quiet_round_misses"] == 0, c
            else:
                misses += c["quiet_round_misses"]
    assert misses > 0, misses


def test_seed_capacity_beyond_four_million_keys(L):
    """the seed order's merge loop never ended once the capacity exceeded 2^22 keys (with one run left its
    second condition stayed true, the host enqueued merges for ever).  Stripes of period 6 at 4K: every pixel of a flank a
    seed of the same magnitude, 2.7 M seeds -- the first lap overflows the frame's initial capacity, the second runs with
    5.5 M slots; then again with lr_set_seed_capacity(6 << 20) on a frame of bars.  Against the oracle."""
    from librectify_amd import synth

    yy, xx = np.mgrid[0:2160, 0:3840].astype(np.float64)
    stripes = (0.5 + 0.4 * np.sin(xx * 2 * np.pi / 6)).astype(np.float32)
    c2 = L.Context(0)
    try:
        c2.set_seed(0)
        ref = O.find_line_segments(stripes, num_threads=8)
        assert ref["n_seeds"] > (2 << 20), ref["n_seeds"]
        c2.stage_filter_host(stripes)
        assert c2.stage_seeds() == ref["n_seeds"]
        c2.stage_flood()
        np.testing.assert_array_equal(c2.download(L.BUF_LABEL), ref["label"])
        _assert_lines_equal(c2.stage_fit(), ref["lines"])
        bars = synth.frame(3840, 2160, 3)
        refb = O.find_line_segments(bars, num_threads=8)
        c2.set_seed_capacity(6 << 20)
        c2.stage_filter_host(bars)
        assert c2.stage_seeds() == refb["n_seeds"]
        c2.stage_flood()
        np.testing.assert_array_equal(c2.download(L.BUF_LABEL), refb["label"])
        _assert_lines_equal(c2.stage_fit(), refb["lines"])
    finally:
        c2.close()


def test_workspace_shrinks_after_a_large_frame(L):
    """VERDICT r04 (weak 11): the reference is stateless, a context keeps a workspace sized by the LARGEST frame it has seen
    (8192 x 8192: gigabytes).  After eight frames in a row of at most a quarter of that size the workspace is given back and
    allocated again at the size in use; lr_context_trim does it at once.  Free device memory (hipMemGetInfo) shows both,
    and the frames that follow a shrink still equal the oracle."""
    import ctypes as C

    from librectify_amd import synth

    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    c2 = L.Context(0)
    try:
        c2.set_seed(0)
        small = synth.frame(1920, 1080, 5)
        ref = O.find_line_segment_groups(small, 19.2, seed=0)[0]
        _assert_lines_equal(c2.find_line_segment_groups(small, 19.2), ref)
        with_small = free_bytes()
        big = np.tile(synth.frame(2048, 2048, 3), (4, 4))
        c2.find_line_segment_groups(big, 80.0)
        with_big = free_bytes()
        assert with_small - with_big > (4 << 30), (with_small, with_big)  # the 8192 x 8192 workspace: more than 4 GB
        for i in range(9):
            _assert_lines_equal(c2.find_line_segment_groups(small, 19.2), ref)
        after = free_bytes()
        assert after - with_big > (4 << 30), (with_big, after)  # given back (the frame slot of the large frame stays until a trim)
        c2.find_line_segment_groups(big, 80.0)
        c2.trim()
        trimmed = free_bytes()
        assert trimmed - with_big > (2 << 30), (with_small, with_big, after, trimmed)  # (what hipMemGetInfo reports lags now and then)
        _assert_lines_equal(c2.find_line_segment_groups(small, 19.2), ref)
        # the lanes of a batch call shrink one by one, in the middle of the call, each only itself (a lane that went through
        # the others' workspaces while they were in the middle of their frames was a GPU fault in bench.py): a batch of 4K
        # frames, then one of sixty small frames through the same five lanes
        c2.set_batch_streams(5)
        big4k = np.stack([synth.frame(3840, 2160, 31 + i) for i in range(2)] * 3)
        c2.find_line_segment_groups_batch_host(big4k, 38.4, capacity=8192, num_threads=4)
        frames = np.stack([synth.frame(960, 540, 100 + i, bars=30) for i in range(6)] * 10)
        out, n, _ = c2.find_line_segment_groups_batch_host(frames, 9.6, capacity=4096, num_threads=4)
        for i in range(6):
            want = O.find_line_segment_groups(frames[i], 9.6, seed=0)[0]
            for j in range(i, 60, 6):
                _assert_lines_equal(out[j, : n[j]], want)
    finally:
        c2.close()


def test_multi_source_rewalks_change_the_time_not_the_labels(L, ctx):
    """Round 4: a seed whose walk was long leaves way-points on its footprint, and its next walk starts from the seed and
    from all of them at once on a team of wavefronts, keeping what is connected to the seed (kernels_flood.hip: team_walk,
    kMulti).  Label image, flood sizes and records with and without, against the oracle; through the hooks that make the
    team's table run out in the middle of such a walk (6: it starts again plainly and moves into a slab; 7: no slab);
    the counter proves that such walks really ran."""
    from librectify_amd import synth

    used = {}
    try:
        for name, img in (("bars", synth.frame(1920, 1080, 7)), ("long", synth.long_bar_frame(1920, 1080, 3, K=24))):
            ref = O.find_line_segments(img)
            for on in (True, False):
                ctx.set_flood_multi_source(on)
                for mode in (1, 6, 7):
                    ctx.set_flood_mode(mode)
                    ctx.stage_filter_host(img)
                    ctx.stage_seeds()
                    ctx.stage_flood()
                    used[(name, on, mode)] = ctx.stage_counters()
                    np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
                    _assert_lines_equal(ctx.stage_fit(), ref["lines"])
    finally:
        ctx.set_flood_multi_source(False)
        ctx.set_flood_mode(1)
    # (way-points are left by walks of a hundred tiles and more: the long bars have them, in every mode)
    assert used[("long", True, 1)]["multi_source_walks"] > 0 and used[("long", False, 1)]["multi_source_walks"] == 0


def test_multi_source_rewalks_from_way_points_on_nearly_every_walk(L):
    """LIBRECTIFY_FLOOD_MULTI_MIN=8 (read once per process: a child process) lets every walk of eight tiles and more leave
    way-points, so that thousands of re-walks per frame start from several sources -- way-points that the footprint has
    lost in the meantime, sources that never meet, tables that run out (modes 6 / 7) included.  Synthetic bars, long bars,
    soft regions and the reference's doc image, all against the oracle."""
    import subprocess
    import sys

    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import oracle_lib as O
import librectify_amd as L
from librectify_amd import synth
g = np.load(os.path.join(%r, "tests", "golden", "doc_image_gray.npy")).astype(np.float32) / np.float32(256.0)
frames = [synth.frame(1920, 1080, 21), synth.frame(960, 540, 4, bars=40), synth.long_bar_frame(1280, 720, 5, K=20),
          synth.region_frame(640, 480, 501), np.ascontiguousarray(g)]
ctx = L.Context(0); ctx.set_seed(0)
walks = 0
for img in frames:
    ref = O.find_line_segments(img)
    for mode in (1, 6, 7):
        ctx.set_flood_mode(mode)
        ctx.stage_filter_host(img); ctx.stage_seeds(); ctx.stage_flood()
        walks += ctx.stage_counters()["multi_source_walks"]
        assert (ctx.download(L.BUF_LABEL) == ref["label"]).all(), "labels differ"
        assert ctx.stage_fit().tobytes() == ref["lines"].tobytes(), "records differ"
assert walks > 1000, walks
print("ok", walks)
""" % (ROOT, ROOT, ROOT)
    env = dict(os.environ, LIBRECTIFY_FLOOD_MULTI_MIN="8", LIBRECTIFY_FLOOD_MULTI="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok"), (r.stdout[-500:], r.stderr[-1500:])


def test_multi_source_rewalks_at_full_size_equal_the_ordered_flood(L, ctx):
    """3840x2160 bench frames 2..4 (frame 1 is test_parallel_flood_equals_ordered_flood_at_full_size): parallel rounds with
    multi-source re-walks against the single-wave ordered kernel."""
    from librectify_amd import synth

    ctx.set_flood_multi_source(True)
    for seed in (2, 3, 4):
        img = synth.frame(3840, 2160, seed)
        out = {}
        for mode in (0, 1):
            ctx.set_flood_mode(mode)
            ctx.stage_filter_host(img)
            ctx.stage_seeds()
            ctx.stage_flood()
            if mode == 1:
                assert ctx.stage_counters()["multi_source_walks"] > 0
            out[mode] = (ctx.download(L.BUF_LABEL), ctx.download(L.BUF_SEED_SIZE))
        ctx.set_flood_mode(1)
        np.testing.assert_array_equal(out[0][0], out[1][0])
        np.testing.assert_array_equal(out[0][1], out[1][1])
    ctx.set_flood_multi_source(False)


def test_multi_device_batch_call_deals_contiguous_blocks(L, ctx):
    """lr_find_line_segment_groups_batch_host_multi (one host thread + stream set per device, contiguous
    blocks of ceil(B / G) frames, results in the caller's arrays, no collective in one process).  On a one-GPU box the
    device list names device 0 two and three times -- independent lane sets, uploaders and pools -- against the
    single-device call and single frames; an odd batch, more list entries than frames, a bad device index."""
    from librectify_amd import synth

    w, h = 640, 480
    frames = np.stack([synth.frame(w, h, 300 + i, bars=20 + i) for i in range(7)])
    ctx.set_seed(0)
    ctx.set_batch_streams(3)
    single = [ctx.find_line_segment_groups(f, 6.4) for f in frames]
    out1, n1, tf1 = ctx.find_line_segment_groups_batch_host(frames, 6.4, num_threads=4, capacity=2048)
    for devs in ([0, 0], [0, 0, 0], [0] * 9):
        out, n, tf = ctx.find_line_segment_groups_batch_host(frames, 6.4, num_threads=4, capacity=2048, devices=devs)
        for b in range(len(frames)):
            assert n[b] == n1[b] == len(single[b])
            _assert_lines_equal(out[b][: n[b]], single[b])
            np.testing.assert_array_equal(tf[b].as_array(), tf1[b].as_array())
    with pytest.raises(L.LibrectifyError):
        ctx.find_line_segment_groups_batch_host(frames, 6.4, capacity=2048, devices=[0, L.device_count()])
    ctx.set_batch_streams(4)


def test_drop_in_thread_context_can_be_released_and_comes_back(L):
    """lr_release_thread_context: the calling thread's drop-in context (workspace, slabs, staging threads) is freed at once;
    the next call through the reference's entry makes a new one and gives the same records."""
    img = FRAMES["640x480"]
    ref, _ = O.find_line_segment_groups(img, 6.4, seed=0)
    for nt in (8, -1):
        _assert_lines_equal(L.find_line_segment_groups(img, 6.4, num_threads=nt), ref)
        L.release_thread_context()
        L.release_thread_context()  # (nothing to release: fine)
    _assert_lines_equal(L.find_line_segment_groups(img, 6.4), ref)


def test_grouping_of_the_reference_golden_lines_on_the_gpu(L, ctx):
    """Pin 4 through the C ABI: the device peeling (kernels_groups.hip) on the reference's own 848 golden lines gives
    the oracle's group ids bit for bit, i.e. the reference's three pencils (tests/test_oracle_pins.py::test_pin4_*)."""
    rows = np.loadtxt(os.path.join(G, "doc_warp_lines.csv"), delimiter=",")
    lines = O.lines_from_rows(rows)
    gold = lines["group_id"].copy()
    blank = lines.copy()
    blank["group_id"] = -1
    for seed in (0, 7):
        ref, _ = O.estimate_line_pencils(blank, seed=seed)
        got = ctx.estimate_line_pencils(blank, seed=seed)
        np.testing.assert_array_equal(got["group_id"], ref["group_id"])
        for g in (0, 1, 2):
            idx = np.nonzero(gold == g)[0]
            assert (got["group_id"][idx] == g).sum() >= 0.94 * len(idx)


def test_direct_estimator_matches_oracle(L, ctx):
    """DirectEstimator (estimator.h:82-96; never instantiated by the reference, so self-golden): solve on a subset, the
    peeling around it, and the whole path with lr_set_estimator(2)."""
    from librectify_amd import synth

    for n, seed in [(120, 3), (1500, 5)]:
        segs = synth.random_segments(n, seed)
        norm, _, _ = O.normalize_lines(segs)
        idx = np.arange(n, dtype=np.int32)[::2]
        np.testing.assert_array_equal(ctx.direct_solve(norm, idx), O.direct_solve(norm, idx))
        np.testing.assert_array_equal(ctx.estimate_line_pencils_direct(segs)["group_id"], O.estimate_line_pencils_direct(segs)["group_id"])
    img = FRAMES["640x480"]
    raw = O.find_line_segments(img, want_label=False)["lines"]
    ref = O.estimate_line_pencils_direct(O.filter_lines(raw, 6.4))
    ctx.set_estimator(2)
    got = ctx.find_line_segment_groups(img, 6.4)
    ctx.set_estimator(0)
    _assert_lines_equal(got, ref)


def test_reference_error_convention_and_strides(L, ctx):
    flat = np.full((64, 80), 0.25, np.float32)
    assert len(L.find_line_segment_groups(flat, 5.0)) == 0  # NULL, *n_lines = 0 (interface.cpp:50-54)
    img = FRAMES["257x131"]
    h, w = img.shape
    ml = float(max(w, h)) / 100.0
    base = L.find_line_segment_groups(img, ml)
    assert len(base) > 0
    padded = np.zeros((h, w + 13), np.float32)
    padded[:, :w] = img
    _assert_lines_equal(L.find_line_segment_groups(padded[:, :w], ml), base)  # stride > width
    # negative stride addresses the same rows from the other end (image.cpp:14-18): no flip
    import ctypes as C

    n = C.c_int(0)
    buf = np.ascontiguousarray(img)
    last_row = buf.ctypes.data + (h - 1) * w * 4
    p = L.lib().find_line_segment_groups(C.c_void_p(last_row), w, h, -w, ml, False, -1, C.byref(n))
    assert p and n.value == len(base)
    got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(n.value * 28,)).view(L.LINE_DTYPE).copy()
    pp = C.c_void_p(p)
    L.lib().release_line_segments(C.byref(pp))
    _assert_lines_equal(got, base)


def test_degenerate_shapes_and_values(L, ctx):
    """Smallest legal frame, one-tile-wide strips, frames below the filter size, and non-finite pixels: the oracle's
    answer where it has one, and in every case a clean return (no hang, no fault, context usable afterwards)."""
    from librectify_amd import synth

    for w, h, seed in [(5, 5, 1), (6, 9, 2), (8, 600, 3), (700, 7, 4), (64, 64, 5)]:
        img = synth.frame(w, h, seed, bars=3)
        ref, _ = O.find_line_segment_groups(img, 2.0, seed=0)
        _assert_lines_equal(ctx.find_line_segment_groups(img, 2.0), ref)
    with pytest.raises(Exception):
        ctx.find_line_segment_groups(np.zeros((4, 40), np.float32), 2.0)  # smaller than the 5x5 filter: an error, said loudly
    img = synth.frame(320, 240, 13, bars=24).copy()
    img[50:60, 70:90] = np.nan
    img[100, 100] = np.inf
    img[150, 200:210] = -np.inf
    got = ctx.find_line_segment_groups(img, 3.2)  # the reference has no defined answer here: only "returns"
    assert got.dtype == L.LINE_DTYPE
    clean = FRAMES["320x240"]
    ref, _ = O.find_line_segment_groups(clean, 3.2, seed=0)
    ctx.set_seed(0)
    _assert_lines_equal(ctx.find_line_segment_groups(clean, 3.2), ref)


def test_random_small_frames_match_the_oracle(L, ctx):
    """Forty seeded frames of random size (5..140 on a side, so most tiles are ragged), bar count, contrast and noise
    level, noiseless ones with exact ties among them: label image and segment records against the oracle."""
    from librectify_amd import synth

    rng = np.random.RandomState(2024)
    for t in range(40):
        w, h = int(rng.randint(5, 141)), int(rng.randint(5, 141))
        img = synth.frame(w, h, 3000 + t, bars=int(rng.randint(1, 9)), noise=[0.0, 0.002, 0.01][t % 3])
        if t % 7 == 0:
            img = np.round(img * 16) / np.float32(16)  # few grey levels: plateaus, equal magnitudes, equal responses
        ref = O.find_line_segments(img.astype(np.float32))
        ctx.stage_filter_host(img.astype(np.float32))
        assert ctx.stage_seeds() == ref["n_seeds"], (t, w, h)
        ctx.stage_flood()
        np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"], err_msg="frame %d (%dx%d)" % (t, w, h))
        _assert_lines_equal(ctx.stage_fit(), ref["lines"])
        if t % 4 == 1:  # every fourth frame through the whole path: filter_lines, grouping, refine on and off
            ml = max(2.0, max(w, h) / 50.0)
            for refine in (False, True):
                full, _ = O.find_line_segment_groups(img.astype(np.float32), ml, refine=refine, seed=0)
                ctx.set_seed(0)
                _assert_lines_equal(ctx.find_line_segment_groups(img.astype(np.float32), ml, refine=refine), full)


def test_batch_entry_point_matches_single_calls(L, ctx):
    """lr_find_line_segment_groups_batch_device keeps several frames in flight; results per frame are those of
    the single-frame call and of the oracle."""
    w, h = 320, 240
    from librectify_amd import synth

    frames = [synth.frame(w, h, 50 + i, bars=20) for i in range(7)]
    d = ctx.device_upload(np.stack(frames))
    ctx.set_seed(0)
    ctx.set_batch_streams(3)
    out, n, tf = ctx.find_line_segment_groups_batch_device(d, w * h, len(frames), w, h, 3.2, capacity=1024)
    ctx.device_free(d)
    for i, f in enumerate(frames):
        ref, _ = O.find_line_segment_groups(f, 3.2, seed=0)
        _assert_lines_equal(out[i][: n[i]], ref)
        Tr = O.transform_to_array(O.compute_rectification_transform(ref, w, h))
        np.testing.assert_array_equal(tf[i].as_array(), Tr)


def _unusual_frames(W, H):
    """Content of the kinds tools/latency_fuzz.py times: noise of several scales, periodic patterns WITHOUT noise (exact ties:
    every pixel of a flank is a seed of the same magnitude), analytic gradients, single edges, glyph-like blocks."""
    from librectify_amd import synth

    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    rng = np.random.RandomState(11)
    blur = synth._gauss_blur
    return {
        "white noise": (0.5 + rng.normal(0, 0.05, (H, W))).astype(np.float32),
        "blurred noise": blur(0.5 + rng.normal(0, 0.2, (H, W)), 3.0).astype(np.float32),
        "checkerboard 16": ((((xx // 16) + (yy // 16)) % 2) * 0.6 + 0.2).astype(np.float32),
        "checkerboard 32 blurred": blur((((xx // 32) + (yy // 32)) % 2) * 0.6 + 0.2, 1.5).astype(np.float32),
        "stripes period 6": (0.5 + 0.4 * np.sin(xx * 2 * np.pi / 6)).astype(np.float32),
        "diagonal stripes": (0.5 + 0.4 * np.sin((xx + 0.5 * yy) * 2 * np.pi / 40)).astype(np.float32),
        "concentric circles": (0.5 + 0.4 * np.sin(np.hypot(xx - W / 2, yy - H / 2) / 12)).astype(np.float32),
        "one vertical edge": blur(np.where(xx > W / 2, 0.8, 0.2), 1.0).astype(np.float32),
        "one slanted edge + noise": (blur(np.where(xx > W / 2 + 0.1 * yy, 0.8, 0.2), 1.0) + rng.normal(0, 0.01, (H, W))).astype(np.float32),
        "radial gradient": (1.0 - np.hypot(xx - W / 2, yy - H / 2) / np.hypot(W / 2, H / 2)).astype(np.float32),
        "radial gradient + noise": (1.0 - np.hypot(xx - W / 2, yy - H / 2) / np.hypot(W / 2, H / 2) + rng.normal(0, 0.003, (H, W))).astype(np.float32),
        "glyph-like blocks": blur((rng.rand(H // 8 + 1, W // 8 + 1) > 0.7).astype(np.float64).repeat(8, 0).repeat(8, 1)[:H, :W] * 0.7 + 0.15, 0.8).astype(np.float32),
        "grid of thin lines": blur(np.where(((xx % 48) < 2) | ((yy % 48) < 2), 0.9, 0.2), 0.7).astype(np.float32),
    }


def test_unusual_content_matches_the_oracle(L, ctx):
    """Round 4: the kinds of content the latency fuzz runs, at 416x304 (tiles ragged on both sides): label image, segment
    records and the grouped result against the oracle -- among them patterns without any noise, where every pixel of a flank
    is a seed with the same magnitude (ties in the seed order, hundreds of seeds per edge with one footprint) and analytic
    gradients whose floods are rings that exhaust the storage tiers (the ordered tail finishes those)."""
    W, H = 416, 304
    ctx.set_seed(0)
    ctx.set_flood_mode(1)
    for name, img in _unusual_frames(W, H).items():
        img = np.ascontiguousarray(img)
        ref = O.find_line_segments(img)
        ctx.stage_filter_host(img)
        assert ctx.stage_seeds() == ref["n_seeds"], name
        ctx.stage_flood()
        np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"], err_msg=name)
        _assert_lines_equal(ctx.stage_fit(), ref["lines"])
        full, _ = O.find_line_segment_groups(img, 4.16, seed=0)
        _assert_lines_equal(ctx.find_line_segment_groups(img, 4.16), full)


def test_lanes_of_a_batch_on_frames_of_every_kind_equal_the_oracle(L, ctx):
    """Round 4: the lanes of a batch call enqueue their flood's later rounds just in time (a look every 20 us), hold giant
    walks back, and carry hints from frame to frame -- here over frames that could not differ more: bars, a ramp without
    edges (held walks), regions, long bars (second tier), in an order that gives every lane a frame of another kind than its
    last.  Every frame's records against the oracle, with two, three and five lanes, and with blind rounds."""
    from librectify_amd import synth

    w, h = 1024, 640
    kinds = [synth.frame(w, h, 81, bars=60), synth.ramp_frame(w, h, 3), _regions(w, h, 12), synth.long_bar_frame(w, h, 5, K=14),
             synth.frame(w, h, 82, bars=25), synth.ramp_frame(w, h, 4), _regions(w, h, 13)]
    order = [0, 1, 2, 3, 4, 5, 6, 1, 0, 3, 2, 5, 4, 6]
    frames = [kinds[i] for i in order]
    refs = [O.find_line_segment_groups(f, 10.24, seed=0)[0] for f in kinds]
    ctx.set_seed(0)
    try:
        for lanes, jit in ((2, True), (3, True), (5, True), (3, False)):
            ctx.set_batch_streams(lanes)
            ctx.set_flood_just_in_time(jit)
            out, n, _ = ctx.find_line_segment_groups_batch_host(frames, 10.24, capacity=8192, num_threads=4)
            for k, i in enumerate(order):
                _assert_lines_equal(out[k][: n[k]], refs[i])
    finally:
        ctx.set_flood_just_in_time(True)
        ctx.set_batch_streams(3)


def test_host_batch_entry_point_pageable_pinned_and_strided(L, ctx):
    """lr_find_line_segment_groups_batch_host (the reference's kind of input, many frames at once): pageable frames
    through page-locked staging buffers, page-locked frames DMA-copied where they lie, both kinds mixed, row strides >
    width, a negative stride, more frames than lanes and than slots of the upload pool, and fewer; all equal to the
    single-frame call."""
    import ctypes as C

    from librectify_amd import synth

    w, h = 333, 190
    frames = np.stack([synth.frame(w, h, 70 + i, bars=18) for i in range(9)])
    ctx.set_seed(0)
    single = [ctx.find_line_segment_groups(f, 3.3) for f in frames]
    tf_single = [L.compute_rectification_transform(s, w, h).as_array() for s in single]
    assert sum(len(s) for s in single) > 100

    def check(out, n, tf, order=range(9)):
        for k, i in enumerate(order):
            _assert_lines_equal(out[k][: n[k]], single[i])
            np.testing.assert_array_equal(tf[k].as_array(), tf_single[i])

    for lanes in (1, 2, 4, 16):
        ctx.set_batch_streams(lanes)
        check(*ctx.find_line_segment_groups_batch_host(frames, 3.3, capacity=1024))
    ctx.set_batch_streams(3)
    check(*ctx.find_line_segment_groups_batch_host([f.copy() for f in frames], 3.3, capacity=1024, num_threads=8))
    padded = np.zeros((9, h, w + 11), np.float32)
    padded[:, :, :w] = frames
    check(*ctx.find_line_segment_groups_batch_host(padded[:, :, :w], 3.3, capacity=1024))  # stride > width
    pinned = ctx.host_alloc((9, h, w))
    pinned[:] = frames
    check(*ctx.find_line_segment_groups_batch_host(pinned, 3.3, capacity=1024))
    check(*ctx.find_line_segment_groups_batch_host(pinned[::2], 3.3, capacity=1024), order=range(0, 9, 2))
    # negative stride through the C entry point: frame pointers address the LAST row (image.cpp:14-18)
    out = np.zeros((9, 1024), L.LINE_DTYPE)
    n = np.zeros(9, np.int32)
    rc = L.lib().lr_find_line_segment_groups_batch_host(ctx._h, C.c_void_p(pinned.ctypes.data + (h - 1) * w * 4), h * w, 9, w, h, -w,
                                                        3.3, 0, -1, out.ctypes.data_as(C.c_void_p), 1024, n.ctypes.data_as(C.c_void_p), None, None)
    assert rc == 0
    for i in range(9):
        _assert_lines_equal(out[i][: n[i]], single[i])
    # page-locked and pageable frames mixed in one call, five times more frames than the upload pool has slots
    # (2 lanes + 6): every slot and staging buffer is reused several times, in whatever order the frames finish
    ctx.set_batch_streams(2)
    order = [(7 * k) % 9 for k in range(40)]
    mixed = [pinned[i] if k % 3 else frames[i].copy() for k, i in enumerate(order)]
    check(*ctx.find_line_segment_groups_batch_host(mixed, 3.3, capacity=1024, num_threads=4), order=order)
    ctx.host_free(pinned)
    with pytest.raises(Exception):
        ctx.find_line_segment_groups_batch_host(np.zeros((3, 4, 40), np.float32), 2.0)  # below the 5x5 filter: loud
    # ... and the context is as good as before after the refused call
    ctx.set_batch_streams(3)
    check(*ctx.find_line_segment_groups_batch_host(frames, 3.3, capacity=1024))


def test_batch_lanes_with_refine_and_with_prosac_and_concurrent_callers(L, ctx):
    """The paths that leave the one-wait pipeline after the line fit (refine = true; the opt-in PROSAC estimator) inside
    batch lanes with prefetching uploads, and the reference's re-entrancy: four host threads calling the drop-in
    find_line_segment_groups at once (a context per thread) -- all against the oracle."""
    import threading

    from librectify_amd import synth

    w, h = 400, 300
    frames = np.stack([synth.frame(w, h, 90 + i, bars=22) for i in range(8)])
    ctx.set_seed(0)
    ctx.set_batch_streams(3)
    refs = [O.find_line_segment_groups(f, 4.0, refine=True, seed=0)[0] for f in frames]
    out, n, _ = ctx.find_line_segment_groups_batch_host(frames, 4.0, refine=True, capacity=1024)
    for i in range(len(frames)):
        _assert_lines_equal(out[i][: n[i]], refs[i])
    ctx.set_estimator(1, 500)
    out, n, _ = ctx.find_line_segment_groups_batch_host(frames, 4.0, capacity=1024)
    ctx.set_estimator(0)
    for i, f in enumerate(frames):
        raw = O.find_line_segments(f, want_label=False)["lines"]
        _assert_lines_equal(out[i][: n[i]], O.estimate_line_pencils_prosac(O.filter_lines(raw, 4.0), T_N=500, seed=0))
    plain = [O.find_line_segment_groups(f, 4.0, seed=0)[0] for f in frames]
    got = [None] * len(frames)

    def call(i):
        got[i] = L.find_line_segment_groups(frames[i], 4.0)  # thread-local context, LIBRECTIFY_SEED default 0

    for rep in range(2):
        ts = [threading.Thread(target=call, args=(i,)) for i in range(rep * 4, rep * 4 + 4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    for i in range(len(frames)):
        _assert_lines_equal(got[i], plain[i])


def test_second_flood_on_a_consumed_filter_output_is_refused(L, ctx):
    """The parallel flood clears the direction mask of the pixels it labels: flooding the same filter output again
    (e.g. after lr_set_flood_mode) must fail loudly instead of returning labels of a mutated mask."""
    img = FRAMES["320x240"]
    ref = O.find_line_segments(img)
    ctx.set_flood_mode(1)
    ctx.stage_filter_host(img)
    ctx.stage_seeds()
    ctx.stage_flood()
    np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
    with pytest.raises(Exception, match="consumed"):
        ctx.stage_flood()
    with pytest.raises(Exception, match="consumed"):
        ctx.download(L.BUF_DMASK)
    _assert_lines_equal(ctx.stage_fit(), ref["lines"])  # the flood's own products are intact
    ctx.stage_filter_host(img)  # and the staged API works again from the filter on
    ctx.stage_seeds()
    ctx.stage_flood()
    np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])


def test_refine_flag_matches_oracle(L, ctx):
    img = FRAMES["320x240"]
    ml = 3.2
    ref, _ = O.find_line_segment_groups(img, ml, refine=True, seed=0)
    ctx.set_seed(0)
    got = ctx.find_line_segment_groups(img, ml, refine=True)
    _assert_lines_equal(got, ref)


def test_many_frames_through_one_context_stay_exact(L, ctx):
    """Workspace reuse (overflow slabs, hash generations, candidate lists) must not leak state from
    one frame into the next: 40 passes over 3 frames with long edges, each equal to its first pass
    and to the oracle."""
    from librectify_amd import synth

    frames = [synth.frame(1280, 720, 31 + i, bars=60) for i in range(3)]
    first = []
    for f in frames:
        ref = O.find_line_segments(f)
        ctx.stage_filter_host(f)
        ctx.stage_seeds()
        ctx.stage_flood()
        np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
        lines = ctx.stage_fit()
        _assert_lines_equal(lines, ref["lines"])
        first.append(lines)
    for rep in range(40):
        i = rep % 3
        ctx.stage_filter_host(frames[i][:, ::-1] if rep % 7 == 3 else frames[i])
        ctx.stage_seeds()
        ctx.stage_flood()
        lines = ctx.stage_fit()
        if rep % 7 != 3:
            _assert_lines_equal(lines, first[i])


def test_parallel_flood_equals_ordered_flood_at_full_size(L, ctx):
    """3840x2160: the round-based parallel flood against the single-wave ordered kernel (exact by
    construction), label image and segment records bit for bit."""
    from librectify_amd import synth

    img = synth.frame(3840, 2160, 1)
    out = {}
    for mode in (0, 1):
        ctx.set_flood_mode(mode)
        ctx.stage_filter_host(img)
        ctx.stage_seeds()
        ctx.stage_flood()
        out[mode] = (ctx.download(L.BUF_LABEL), ctx.download(L.BUF_SEED_SIZE), ctx.stage_fit())
    ctx.set_flood_mode(1)
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    _assert_lines_equal(out[0][2], out[1][2])


def test_staged_rounds_of_a_batch_equal_the_single_frame_call_at_full_size(L, ctx):
    """3840x2160 through the batch entry point with the staged start of the flood rounds switched on (the rounds begin
    on the strongest eighth of the seeds and widen the window round by round) against the single-frame call (all
    seeds from round one), which the test above ties to the ordered flood: same segments, same groups."""
    from librectify_amd import synth

    w, h = 3840, 2160
    frames = [synth.frame(w, h, 1), synth.frame(w, h, 2)]
    ctx.set_seed(0)
    single = [ctx.find_line_segment_groups(f, max(w, h) / 100.0) for f in frames]
    d = ctx.device_upload(np.stack(frames))
    ctx.set_batch_streams(2)
    ctx.set_flood_staged(True)
    out, n, tf = ctx.find_line_segment_groups_batch_device(d, w * h, len(frames), w, h, max(w, h) / 100.0, capacity=4096)
    ctx.set_flood_staged(False)
    ctx.device_free(d)
    for i in range(len(frames)):
        assert n[i] == len(single[i]) and n[i] > 500
        _assert_lines_equal(out[i][: n[i]], single[i])


def _long_bars(W, H, seed, K=14):
    """Bars across most of the frame: edges of well over 1500 px, i.e. walks that outgrow the first storage tier."""
    from librectify_amd import synth

    rng = np.random.RandomState(seed)
    img = np.full((H, W), 0.5, np.float64)
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(K):
        c = np.array([rng.uniform(0.4, 0.6) * W, rng.uniform(0.1, 0.9) * H])
        ang = rng.uniform(-0.08, 0.08)
        d = np.array([np.cos(ang), np.sin(ang)])
        nrm = np.array([-d[1], d[0]])
        px, py = xx - c[0], yy - c[1]
        m = (np.abs(px * d[0] + py * d[1]) <= rng.uniform(0.6, 0.95) * W / 2) & (np.abs(px * nrm[0] + py * nrm[1]) <= rng.uniform(3, 9))
        img[m] += rng.uniform(0.1, 0.4) * (1 if rng.rand() < 0.5 else -1)
    img = synth._gauss_blur(np.clip(img, 0, 1), 1.0) + rng.normal(0, 0.005, size=img.shape)
    return img.astype(np.float32)


def test_every_storage_tier_of_the_flood_is_exact_on_long_edges(L, ctx):
    """2560x480 with edges of 1500-2400 px against the oracle: default (walks restart in the second LDS tier), mode 4
    (no second tier: they carry on in global slabs, and when the pool runs out the ordered tail finishes), mode 3 (two
    slabs), mode 2 (no slab at all), modes 6 and 7 (the second tier's team of wavefronts runs out of storage in the middle
    of a level: the team moves into a global slab with the unprocessed records / no slab: incomplete walk, barrier,
    ordered tail).  The counters prove that each path was really taken."""
    img = _long_bars(2560, 480, 5)
    ref = O.find_line_segments(img)
    used = {}
    for mode in (1, 4, 3, 2, 5, 6, 7):
        ctx.set_flood_mode(mode)
        ctx.stage_filter_host(img)
        ctx.stage_seeds()
        ctx.stage_flood()
        used[mode] = ctx.stage_counters()
        np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
        _assert_lines_equal(ctx.stage_fit(), ref["lines"])
    ctx.set_flood_mode(1)
    assert used[1]["second_tier_seeds"] > 0 and used[1]["ordered_tail_seeds"] == 0
    assert used[4]["second_tier_seeds"] == 0 and used[4]["slabs"] > 2
    assert used[3]["ordered_tail_seeds"] > 0
    assert used[2]["slabs"] == 0 and used[2]["ordered_tail_seeds"] > 0
    assert used[5]["second_tier_seeds"] > 0 and used[5]["slabs"] == 0 and used[5]["ordered_tail_seeds"] > 0
    assert used[6]["second_tier_seeds"] > 0 and used[6]["slabs"] > 0  # (the pool may run out as well: then the tail finishes)
    assert used[7]["second_tier_seeds"] > 0 and used[7]["slabs"] == 0 and used[7]["ordered_tail_seeds"] > 0


def _regions(W, H, seed):
    """Soft blobs and ramps: floods that are regions (wide frontiers), the kind the second tier's team counts as wide."""
    from librectify_amd import synth

    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.full((H, W), 0.4, np.float64)
    for _ in range(10):
        cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(40, 160)
        img += rng.uniform(0.1, 0.3) * np.exp(-(((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r)))
    img += 0.15 * xx / W
    img = synth._gauss_blur(np.clip(img, 0, 1), 2.0) + rng.normal(0, 0.003, size=img.shape)
    return img.astype(np.float32)


def test_what_a_context_carries_from_frame_to_frame_never_changes_a_result(L):
    """Frames of lines, of long bars and of regions in turn on ONE fresh context: second tier, hold-back from the start and
    the early hand-over on frames of regions are decided from the previous frame and from the frame's own rounds; every
    frame must equal the oracle whatever came before it (round 3: the rules changed; the results may not)."""
    c = L.Context(0)
    c.set_seed(0)
    frames = [FRAMES["640x480"], _long_bars(1600, 400, 9), _regions(960, 540, 3), FRAMES["320x240"], _regions(960, 540, 4),
              _long_bars(1600, 400, 10), _regions(960, 540, 3)]
    order = [0, 1, 2, 3, 2, 2, 0, 4, 1, 5, 6, 0]
    refs = {}
    for i in order:
        if i not in refs:
            refs[i] = O.find_line_segment_groups(frames[i], max(frames[i].shape) / 100.0, seed=0)[0]
        got = c.find_line_segment_groups(frames[i], max(frames[i].shape) / 100.0)
        _assert_lines_equal(got, refs[i])
    c.close()


def test_a_round_whose_list_outgrows_its_grid_is_exact(L):
    """From the fifth round on the exploration has no `rest` launch behind it: a list longer than the grid is walked in its
    first entries only, and the survivors pass before it has put the round's barrier at the lowest seed behind them.
    LIBRECTIFY_FLOOD_TEST_GRID=1 makes every such round with more than one seed left outgrow its grid (more rounds, or the
    flood stalls and the ordered tail finishes it: slow, and exact).  That is the life of rounds enqueued BLINDLY (LIBRECTIFY_FLOOD_JIT=0 here); a round
    enqueued just in time knows its list's length and brings the `rest` launch when the grid is too small (round 4: second
    child -- same frames, same records, and no seed left to the ordered tail).  The knobs are read once per process."""
    import subprocess
    import sys

    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import oracle_lib as O
import librectify_amd as L
from librectify_amd import synth
ctx = L.Context(0); ctx.set_seed(0)
tails = 0
rounds = 0
for img in (synth.frame(3840, 2160, 1), synth.frame(1920, 1080, 9), synth.frame(960, 540, 4, bars=40)):
    ref, _ = O.find_line_segment_groups(img, max(img.shape) / 100.0, seed=0)
    got = ctx.find_line_segment_groups(img, max(img.shape) / 100.0)
    assert got.tobytes() == ref.tobytes(), "mismatch"
    tails += ctx.stage_counters()["ordered_tail_seeds"]
    rounds += ctx.stage_counters()["flood_rounds"]
print("ok", tails, rounds)
""" % (ROOT, ROOT)
    tails, rounds = {}, {}
    for jit in ("0", "1"):
        env = dict(os.environ, LIBRECTIFY_FLOOD_TEST_GRID="1", LIBRECTIFY_FLOOD_JIT=jit)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().startswith("ok"), (jit, r.stdout[-500:], r.stderr[-1500:])
        tails[jit], rounds[jit] = int(r.stdout.split()[1]), int(r.stdout.split()[2])
    # blind: one entry a round is walked from the fifth round on, the rest waits behind the barrier (or the flood stalls
    # and the ordered tail finishes it); just in time: the launch for the rest is there, no round more than without the knob
    assert tails["0"] > 0 or rounds["0"] > rounds["1"], ("the knob did not bite", tails, rounds)
    assert tails["1"] == 0, tails  # (a frame enqueues at most four rounds blindly: those that have the launch anyway)


def test_component_sort_classes_up_to_a_flood_of_20000_pixels(L, ctx):
    """The per-component pixel sort has four size classes (<= 64 px in a wavefront, <= 4096 and <= 16384 in LDS, beyond
    that in global memory).  A 3200x200 frame with a soft horizontal step (a 6-px wide, 3200-px long flood) and bars of
    graded length exercises all of them; label image and segment records against the oracle."""
    from librectify_amd import synth

    rng = np.random.RandomState(77)
    W, H = 3200, 200
    img = np.full((H, W), 0.3, np.float64)
    img[100:, :] += 0.4                       # one step across the whole width ...
    img = synth._gauss_blur(img, 4.0)         # ... made soft: its flood is ~6 px thick
    for k, (x0, ln) in enumerate([(50, 40), (200, 300), (700, 900), (1700, 1400)]):  # sharper bars of graded length
        img[20 + 3 * k: 26 + 3 * k, x0: x0 + ln] += 0.25
    img[60:66, 100:3000] += 0.25
    img = synth._gauss_blur(img, 1.0) + rng.normal(0, 0.002, size=img.shape)
    img = img.astype(np.float32)
    ref = O.find_line_segments(img)
    sizes = np.bincount(ref["label"][ref["label"] >= 0])
    assert sizes.max() > 16384 and ((sizes > 4096) & (sizes <= 16384)).any() and ((sizes > 64) & (sizes <= 4096)).any()
    ctx.set_flood_mode(1)
    ctx.stage_filter_host(img)
    assert ctx.stage_seeds() == ref["n_seeds"]
    ctx.stage_flood()
    np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
    _assert_lines_equal(ctx.stage_fit(), ref["lines"])


def test_natural_image_at_4k_matches_the_oracle(L, ctx):
    """The reference's doc image upsampled to 3840x2160 (cubic spline): 53 760 seeds, walks of over a thousand tiles,
    second storage tier and the hold-back of the weakest seeds all in play; full path against the oracle."""
    import scipy.ndimage as ndi

    g = np.load(os.path.join(G, "doc_image_gray.npy")).astype(np.float32) / np.float32(256.0)
    w, h = 3840, 2160
    img = np.ascontiguousarray(ndi.zoom(g, (h / g.shape[0], w / g.shape[1]), order=3).astype(np.float32)[:h, :w])
    assert img.shape == (h, w)
    ctx.set_seed(0)
    ctx.set_flood_mode(1)  # also forgets what the previous frame's walks needed: the second tier is on from round one
    got = ctx.find_line_segment_groups(img, max(w, h) / 100.0)
    used = ctx.stage_counters()
    ref, _ = O.find_line_segment_groups(img, max(w, h) / 100.0, seed=0)
    _assert_lines_equal(got, ref)
    assert len(got) > 300 and used["second_tier_seeds"] > 0 and used["ordered_tail_seeds"] == 0
    # a frame without long walks in between: the next flood still has the second tier from its first round (round 2 kept
    # it only after a frame that had used it, and this frame's long walks then went to global slabs: 6.4 ms instead of 1.7)
    ctx.find_line_segment_groups(FRAMES["320x240"], 3.2)
    got2 = ctx.find_line_segment_groups(img, max(w, h) / 100.0)
    _assert_lines_equal(got2, ref)
    used = ctx.stage_counters()
    assert used["second_tier_seeds"] > 0 and used["slabs"] == 0


def _pencil(vp, n_on, n_off, seed):
    rng = np.random.RandomState(seed)
    rows = []
    for i in range(n_on + n_off):
        c = rng.uniform(50, 950, 2)
        if i < n_on:
            d = np.array(vp) - c
            d /= np.linalg.norm(d)
            a = rng.normal(0, 0.002)
            d = np.array([d[0] * np.cos(a) - d[1] * np.sin(a), d[0] * np.sin(a) + d[1] * np.cos(a)])
        else:
            t = rng.uniform(0, np.pi)
            d = np.array([np.cos(t), np.sin(t)])
        ln = rng.uniform(30, 120)
        p1, p2 = c - d * ln / 2, c + d * ln / 2
        rows.append([p1[0], p1[1], p2[0], p2[1], 1, 0, -1])
    return O.lines_from_rows(np.array(rows))


def test_diamond_space_accumulator(L, ctx):
    """Opt-in cascaded-Hough estimator (reference cht.h:13-24; its cht.cpp is an uncompiled sketch, so parity is
    unpinned): the LDS accumulator equals the oracle's cell for cell, and known vanishing points of synthetic
    pencils are recovered to the accumulator's angular resolution."""
    centre = np.array([500.0, 500.0])
    for vp, seed in [((1800.0, 420.0), 1), ((-900.0, 300.0), 2), ((520.0, -4000.0), 3), ((500.0, 380.0), 4)]:
        for n_on, n_off in [(260, 140), (3000, 6000)]:  # the second case spans several workgroups
            lines = _pencil(vp, n_on, n_off, seed)
            ref_vp, ref_acc = O.cht_vanishing_point(lines, 128)
            got_vp, got_acc = ctx.cht_vanishing_point(lines, 128)
            np.testing.assert_array_equal(got_acc, ref_acc)
            np.testing.assert_array_equal(got_vp, ref_vp)
            a = np.array(vp) - centre
            b = got_vp[:2] - centre if got_vp[2] != 0 else got_vp[:2]
            cosang = abs(np.dot(a, b)) / (np.linalg.norm(a) * np.linalg.norm(b))
            near = got_vp[2] != 0 and np.linalg.norm(got_vp[:2] - np.array(vp)) < 25.0  # a finite VP close to the frame
            assert near or cosang > np.cos(np.radians(3.0)), (vp, got_vp)


def test_refine_pair_kernel_matches_oracle(L, ctx):
    """postprocess_lines_segments on raw detector output: small n takes the host loop, n >= 2048 the GPU pair
    kernel; both against the oracle (self-golden: the reference pins nothing for refine)."""
    from librectify_amd import synth

    for w, h, seed in [(320, 240, 13), (1920, 1080, 1000)]:
        raw = O.find_line_segments(synth.frame(w, h, seed), want_label=False)["lines"]
        assert (len(raw) >= 2048) == (w > 1000)
        ref = O.refine_lines(raw)
        got = ctx.refine_lines(raw)
        _assert_lines_equal(got, ref)
        assert len(got) < len(raw)


def test_size_independent_properties_at_full_size(L, ctx):
    """3840x2160 (BASELINE configs[1..2]): properties that need no oracle run."""
    from librectify_amd import synth

    img = synth.frame(3840, 2160, 1)
    h, w = img.shape
    ctx.stage_filter_host(img)
    n_seeds = ctx.stage_seeds()
    idx = ctx.download(L.BUF_SEED_IDX)
    thr = ctx.download(L.BUF_SEED_THR)
    dx, dy = ctx.download(L.BUF_DX), ctx.download(L.BUF_DY)
    mag = np.sqrt(dx * dx + dy * dy)
    assert (dx[:2] == 0).all() and (dx[-2:] == 0).all() and (dx[:, :2] == 0).all() and (dx[:, -2:] == 0).all()
    sm = mag.reshape(-1)[idx]
    assert (np.diff(sm) <= 0).all()  # sortedness
    assert len(np.unique(idx)) == n_seeds
    assert (sm > mag.max() * np.float32(1 - np.float32(0.95))).all()
    ctx.stage_flood()
    lab = ctx.download(L.BUF_LABEL)
    sizes = ctx.download(L.BUF_SEED_SIZE)
    counts = np.bincount(lab[lab >= 0], minlength=n_seeds)
    np.testing.assert_array_equal(counts, sizes)  # checksum of the label image against per-seed sizes
    started = sizes > 0
    assert (lab.reshape(-1)[idx[started]] == np.nonzero(started)[0]).all()  # a started seed owns its own pixel
    assert (lab.reshape(-1)[idx[~started]] >= 0).all()  # a skipped seed was claimed by an earlier one
    assert (lab.reshape(-1)[idx[~started]] < np.nonzero(~started)[0]).all()
    assert (lab[0] == -1).all() and (lab[-1] == -1).all() and (lab[:, 0] == -1).all() and (lab[:, -1] == -1).all()
    lines = ctx.stage_fit()
    assert len(lines) == (sizes > 5).sum()
    assert np.isfinite(np.stack([lines[k] for k in ("x1", "y1", "x2", "y2", "weight", "err")])).all()
    # idempotence: the same frame again gives the same records
    ctx.stage_filter_host(img)
    ctx.stage_seeds()
    ctx.stage_flood()
    _assert_lines_equal(ctx.stage_fit(), lines)
