"""BASELINE.json configs[1..4] at FULL size against the CPU oracle, through the C ABI.

configs[1], [2]: the bench's own 3840x2160 frame (seed 1): every stage product, then the whole path + transform;
configs[3]:      1920x1080 frames, seeds 1000+i, through the batch entry points (device-resident and host-pointer);
configs[4]:      8192x8192 tiled frame (seed 7, 512-px blocks), ~24k segments, RANSAC and PROSAC with T_N = 100 000.
Bars: records bit-exact (endpoints, weight, err, group id); vanishing points within 1e-4 on unit-normalised
homogeneous coordinates.  The oracle runs threaded here (its results do not depend on the thread count:
tests/test_oracle_pins.py::test_oracle_threaded_equals_serial)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


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


def _assert_lines_equal(a, b):
    assert len(a) == len(b), (len(a), len(b))
    if len(a):
        av = np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint32).reshape(len(a), 7)
        bv = np.frombuffer(np.ascontiguousarray(b).tobytes(), np.uint32).reshape(len(b), 7)
        bad = np.nonzero((av != bv).any(axis=1))[0]
        assert len(bad) == 0, "%d mismatching records, first at %d: %s vs %s" % (len(bad), bad[0], a[bad[0]], b[bad[0]])


def _unit(v):
    v = np.asarray(v, np.float64)
    return v / np.linalg.norm(v)


def _assert_transform_close(Tg, Tr):
    for k in (4, 5):  # vanishing points: 1e-4 on unit-normalised homogeneous coordinates (north_star)
        assert np.abs(_unit(Tg[k]) - _unit(Tr[k])).max() < 1e-4
    np.testing.assert_allclose(Tg[:4], Tr[:4], rtol=1e-4, atol=1e-3)


def test_config_2_3_bench_frame_4k_full_path_matches_oracle(L, ctx):
    """frame(3840, 2160, 1) — the frame bench.py times: stage products and the whole path against the oracle."""
    from librectify_amd import synth

    w, h = 3840, 2160
    img = synth.frame(w, h, 1)
    T = O.max_threads()
    ref = O.find_line_segments(img, num_threads=T)
    ctx.set_flood_mode(1)
    ctx.stage_filter_host(img)
    assert ctx.stage_seeds() == ref["n_seeds"]
    ctx.stage_flood()
    np.testing.assert_array_equal(ctx.download(L.BUF_LABEL), ref["label"])
    _assert_lines_equal(ctx.stage_fit(), ref["lines"])
    ml = max(w, h) / 100.0
    full, _ = O.find_line_segment_groups(img, ml, seed=0, num_threads=T)
    ctx.set_seed(0)
    got = ctx.find_line_segment_groups(img, ml)
    _assert_lines_equal(got, full)
    assert len(got) > 500 and set(got["group_id"].tolist()) >= {0, 1, 2}
    _assert_transform_close(L.compute_rectification_transform(got, w, h).as_array(),
                            O.transform_to_array(O.compute_rectification_transform(full, w, h)))
    # the reference's own entry point (host pointer, thread-local context) gives the same records
    _assert_lines_equal(L.find_line_segment_groups(img, ml), full)


def test_config_4_batch_of_1080p_frames_matches_oracle(L, ctx):
    """frame(1920, 1080, 1000 + i), i < 16, through both batch entry points (frames in flight on several lanes)."""
    from librectify_amd import synth

    w, h, B = 1920, 1080, 16
    ml = max(w, h) / 100.0
    T = O.max_threads()
    frames = np.stack([synth.frame(w, h, 1000 + i) for i in range(B)])
    refs = [O.find_line_segment_groups(frames[i], ml, seed=0, num_threads=T)[0] for i in range(B)]
    ctx.set_seed(0)
    ctx.set_batch_streams(6)
    d = ctx.device_upload(frames)
    out, n, tf = ctx.find_line_segment_groups_batch_device(d, w * h, B, w, h, ml, capacity=4096)
    ctx.device_free(d)
    out_h, n_h, tf_h = ctx.find_line_segment_groups_batch_host(frames, ml, capacity=4096)
    for i in range(B):
        assert n[i] > 200
        _assert_lines_equal(out[i][: n[i]], refs[i])
        _assert_lines_equal(out_h[i][: n_h[i]], refs[i])
        Tr = O.transform_to_array(O.compute_rectification_transform(refs[i], w, h))
        np.testing.assert_array_equal(tf[i].as_array(), Tr)
        np.testing.assert_array_equal(tf_h[i].as_array(), Tr)


def test_config_4_all_512_frames_same_results_from_pageable_pinned_device_and_two_lane_sets(L, ctx):
    """BASELINE configs[3] at its full size: 512 frames 1920x1080 in ONE call (the pool of device frames is recycled some
    forty times, the lanes take 85 frames each).  The sixteen distinct frames are tied to the oracle by the test above;
    here every one of the 512 results must equal its base frame's (or its flip's own single-call result), whichever way the
    frames come in: pageable host memory, page-locked host memory, HBM, and the multi-device call with two lane sets."""
    from librectify_amd import synth

    w, h, B = 1920, 1080, 512
    ml = max(w, h) / 100.0
    base = [synth.frame(w, h, 1000 + i) for i in range(8)]
    variants = []
    for b in base:
        variants += [b, b[:, ::-1], b[::-1, :], b[::-1, ::-1]]
    frames = np.empty((B, h, w), np.float32)
    for i in range(B):
        frames[i] = variants[i % len(variants)]
    ctx.set_seed(0)
    ctx.set_batch_streams(6)
    single = [ctx.find_line_segment_groups(np.ascontiguousarray(v), ml) for v in variants]
    out = np.zeros((B, 2048), L.LINE_DTYPE)

    def check(res):
        o, n, tf = res
        for i in range(B):
            s = single[i % len(variants)]
            assert n[i] == len(s), (i, n[i], len(s))
            assert o[i][: n[i]].tobytes() == s.tobytes(), i

    check(ctx.find_line_segment_groups_batch_host(frames, ml, num_threads=8, capacity=2048, out=out))
    pinned = ctx.host_alloc((B, h, w))
    pinned[:] = frames
    check(ctx.find_line_segment_groups_batch_host(pinned, ml, num_threads=8, capacity=2048, out=out))
    check(ctx.find_line_segment_groups_batch_host(frames, ml, num_threads=8, capacity=2048, out=out, devices=[0, 0]))
    ctx.host_free(pinned)
    d = ctx.device_upload(frames)
    check(ctx.find_line_segment_groups_batch_device(d, w * h, B, w, h, ml, capacity=2048, out=out))
    ctx.device_free(d)


def test_config_5_8k_tiled_frame_ransac_and_prosac_100k_match_oracle(L, ctx):
    """frame(8192, 8192, 7, 512-px blocks): 326 590 seeds, 162 186 components, 24 007 segments — the sizes where the
    29-bit index packing, the candidate lists and 32-bit offsets are closest to their limits.  Default RANSAC and
    PROSAC with T_N = 100 000 (the reference's default T_N is 9, prosac.h:116; BASELINE configs[4] raises it)."""
    from librectify_amd import synth

    w = h = 8192
    img = synth.frame(w, h, 7, bars=6000, tile=512)
    T = O.max_threads()
    det = O.find_line_segments(img, num_threads=T, want_label=False)
    filt = O.filter_lines(det["lines"], 20.0)
    assert len(filt) > 20000
    ctx.set_flood_mode(1)
    ctx.set_seed(0)
    ctx.set_estimator(0)
    got = ctx.find_line_segment_groups(img, 20.0, capacity=200000)
    used = ctx.stage_counters()
    assert used["seeds"] == det["n_seeds"] and used["components"] == len(det["lines"])
    assert used["ordered_tail_seeds"] == 0  # the parallel rounds finished the frame (no storage exhaustion)
    ref_r, _ = O.estimate_line_pencils(filt, seed=0, num_threads=T)
    _assert_lines_equal(got, ref_r)
    ctx.set_estimator(1, 100000)
    got_p = ctx.find_line_segment_groups(img, 20.0, capacity=200000)
    ctx.set_estimator(0)
    ref_p = O.estimate_line_pencils_prosac(filt, T_N=100000, seed=0)
    _assert_lines_equal(got_p, ref_p)
    assert set(got_p["group_id"].tolist()) == {-1, 0, 1, 2, 3}
    _assert_transform_close(L.compute_rectification_transform(got_p, w, h).as_array(),
                            O.transform_to_array(O.compute_rectification_transform(ref_p, w, h)))
