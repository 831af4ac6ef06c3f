"""What a library behind the reference's stateless API (src/librectify.h:111-123) owes its host process besides
results: contexts that fail cleanly when memory runs out, calls that fail cleanly and recover, helper threads that
sleep between calls."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import librectify_amd as L

    L.lib()
    assert L.device_count() > 0, "GPU tests need a GPU"
    return L


_CREATE_SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import librectify_amd as L
try:
    c = L.Context(0)
except L.LibrectifyError as e:
    print("ERR", e)
    sys.exit(0)
print("CREATED")
c.close()
"""


@pytest.mark.parametrize("nth", [1, 17, 20, 24, 28, 30])
def test_context_creation_fails_cleanly_when_an_allocation_fails(L, nth):
    """lr_context_create makes 30 events, device words and page-locked words; any of them failing must give an error
    and no context (round 3 ignored the results and faulted in the first kernel).  The hook is read once per process."""
    env = dict(os.environ, LIBRECTIFY_TEST_CREATE_FAIL=str(nth))
    r = subprocess.run([sys.executable, "-c", _CREATE_SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ERR" in r.stdout and "lr_context_create" in r.stdout and "CREATED" not in r.stdout, r.stdout


def test_a_flood_workspace_that_does_not_fit_fails_the_call_and_the_next_call_recovers(L):
    from librectify_amd import synth

    img = synth.frame(320, 240, 13, bars=24)
    ref, _ = O.find_line_segment_groups(img, 3.2, seed=0)
    c = L.Context(0)
    try:
        c.set_seed(0)
        os.environ["LIBRECTIFY_FLOOD_SLABS"] = "2000000"  # x 2.25 MB: more than the card has
        try:
            with pytest.raises(L.LibrectifyError) as ei:
                c.find_line_segment_groups(img, 3.2)
            assert "hipMalloc" in str(ei.value)
        finally:
            del os.environ["LIBRECTIFY_FLOOD_SLABS"]
        got = c.find_line_segment_groups(img, 3.2)
        assert got.tobytes() == ref.tobytes()
    finally:
        c.close()


def test_staging_helpers_sleep_between_calls(L):
    """num_threads = 8 starts the context's staging helpers; they stay with the context from call to call and must cost
    an idle process nothing (round 3: eight threads polling every 20 us)."""
    from librectify_amd import synth

    img = synth.frame(1920, 1080, 3)  # pageable: goes through the staging copy
    c = L.Context(0)
    try:
        c.set_seed(0)
        a = c.find_line_segment_groups(img, 19.2, num_threads=8)
        time.sleep(0.05)  # (past the helpers' spin phase)
        t0 = time.process_time()
        time.sleep(0.5)
        cpu = time.process_time() - t0
        assert cpu < 0.02, "helpers burnt %.3f s of CPU in 0.5 s of idle time" % cpu
        b = c.find_line_segment_groups(img, 19.2, num_threads=8)  # ... and wake up for the next frame
        assert a.tobytes() == b.tobytes() and len(a) > 0
    finally:
        c.close()


def test_pageable_frames_page_locked_where_they_lie_give_the_same_results(L):
    """LIBRECTIFY_REGISTER_FRAMES=<threads> (read once per process: a child): the batch call pins the caller's pageable
    frames in place (hipHostRegister) a few frames ahead of the uploader, sends them by DMA from there -- no staging copy,
    one pass through host DRAM instead of three -- and unpins them when the call is over.  Same records as single calls,
    twice over the same buffers (they must have come back pageable), one of them a view with a row stride."""
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import oracle_lib as O
import librectify_amd as L
from librectify_amd import synth
ctx = L.Context(0); ctx.set_seed(0); ctx.set_batch_streams(4)
wide = np.zeros((12, 540, 1000), np.float32)
frames = wide[:, :, :960]
for i in range(12): frames[i] = synth.frame(960, 540, 50 + i %% 4, bars=30)
want = [O.find_line_segment_groups(np.ascontiguousarray(frames[i]), 9.6, seed=0)[0] for i in range(4)]
for rep in range(2):
    out, n, _ = ctx.find_line_segment_groups_batch_host(frames, 9.6, capacity=4096, num_threads=4)
    for i in range(12):
        assert out[i, : n[i]].tobytes() == want[i %% 4].tobytes(), ("mismatch", rep, i)
print("ok")
""" % (ROOT, ROOT)
    env = dict(os.environ, LIBRECTIFY_REGISTER_FRAMES="3")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-1500:])


def test_round5_launch_savings_and_host_path_defaults_against_their_comparison_knobs(L):
    """Round 5's launch savings and the registration of pageable frames are defaults; each has a knob that gives the old
    path back (read once per process: children).  With every knob at its comparison setting -- three launches for the seed
    selection, a copy of the flood's control block, every blind round with the second tier, the staging copy for every
    frame, bands of 1 MB for a page-locked single frame -- and with none set: the same records as the oracle's, from single
    calls out of pageable and page-locked buffers and from a batch that holds one frame twice."""
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import oracle_lib as O
import librectify_amd as L
from librectify_amd import synth
ctx = L.Context(0); ctx.set_seed(0); ctx.set_batch_streams(3)
kinds = [synth.frame(960, 540, 60, bars=30), synth.region_frame(960, 540, 500), synth.frame(960, 540, 61, bars=45)]
want = [O.find_line_segment_groups(f, 9.6, seed=0)[0] for f in kinds]
pinned = ctx.host_alloc((3, 540, 960))
for i in range(3): pinned[i] = kinds[i]
for rep in range(2):
    for i in (0, 1, 2, 0):
        assert ctx.find_line_segment_groups(kinds[i], 9.6).tobytes() == want[i].tobytes(), ("pageable", rep, i)
        assert ctx.find_line_segment_groups(pinned[i], 9.6).tobytes() == want[i].tobytes(), ("page-locked", rep, i)
    order = [0, 1, 2, 1, 0, 2, 2, 1]
    out, n, _ = ctx.find_line_segment_groups_batch_host([kinds[i] for i in order], 9.6, capacity=4096, num_threads=4)
    for j, i in enumerate(order):
        assert out[j, : n[j]].tobytes() == want[i].tobytes(), ("batch", rep, j)
print("ok")
""" % (ROOT, ROOT)
    knobs = dict(LIBRECTIFY_SEED_SELECT_FUSED="0", LIBRECTIFY_FLOOD_MIRROR="0", LIBRECTIFY_FLOOD_CALM_HINT="0",
                 LIBRECTIFY_REGISTER_FRAMES="0", LIBRECTIFY_UPLOAD_BAND_KB="1024")
    for extra in ({}, knobs):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (extra, r.stdout[-500:], r.stderr[-1500:])
