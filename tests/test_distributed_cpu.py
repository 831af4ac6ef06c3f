"""world_size-2 gloo test of the N>1 path: shard frames by rank, process locally, gather once.
The per-frame work is done by the CPU oracle here (no GPU in this container); on GPUs bench.py runs
the same shard/gather code around the HIP path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import oracle_lib as O
    from librectify_amd import distributed as D
    from librectify_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = D.shard_range(n_frames, rank, world)
    lines, tfs = [], []
    for i in range(b, e):
        img = synth.frame(96, 64, 100 + i, bars=8)
        l, _ = O.find_line_segment_groups(img, 2.0, seed=0)
        lines.append(l)
        tfs.append(O.transform_to_array(O.compute_rectification_transform(l, 96, 64)))
    all_lines, all_tf = D.gather_results(lines, np.array(tfs, np.float32).reshape(-1, 6, 3), n_frames)
    q.put((rank, [x.tobytes() for x in all_lines], all_tf.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [4, 5])
def test_shard_and_gather_world2(n_frames):
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from librectify_amd import synth

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_frames
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process answer
    exp_lines, exp_tf = [], []
    for i in range(n_frames):
        img = synth.frame(96, 64, 100 + i, bars=8)
        l, _ = O.find_line_segment_groups(img, 2.0, seed=0)
        exp_lines.append(l.tobytes())
        exp_tf.append(O.transform_to_array(O.compute_rectification_transform(l, 96, 64)))
    exp_tf = np.array(exp_tf, np.float32).tobytes()
    for rank, lines, tf in res:
        assert lines == exp_lines, "rank %d gathered different segments" % rank
        assert tf == exp_tf


def test_shard_range_covers_everything():
    from librectify_amd import distributed as D

    for n in (0, 1, 7, 8, 512):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                b, e = D.shard_range(n, r, world)
                got += list(range(b, e))
            assert got == list(range(n))


def _worker_async(rank, world, port, n_steps, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from librectify_amd import LINE_DTYPE
    from librectify_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gw = D.GatherWorker(keep=True)
    n_total = 5
    b, e = D.shard_range(n_total, rank, world)
    buf = np.zeros((e - b, 7), LINE_DTYPE)  # (a buffer the "next step" overwrites: the worker gets copies)
    for step in range(n_steps):
        lines = []
        for i in range(b, e):
            n = (i + step) % 7
            buf[i - b, :n]["x1"] = np.arange(n, dtype=np.float32) + 100 * i + step
            buf[i - b, :n]["group_id"] = i
            lines.append(buf[i - b, :n].copy())
        tf = np.full((e - b, 6, 3), float(step), np.float32)
        gw.submit(lines, tf, n_total)
        buf[:] = 0  # the rank goes on while the gather runs
    gw.drain()
    all_lines, all_tf = gw.results
    dist.barrier()  # (a collective of the main thread: only after the drain)
    q.put((rank, [x.tobytes() for x in all_lines], all_tf.tobytes()))
    gw.close()
    dist.destroy_process_group()


def test_gather_worker_gathers_in_step_order_beside_the_ranks_work():
    """bench.py with a process group hands each step's results to librectify_amd.distributed.GatherWorker: a thread per rank that
    issues the rank's collectives in step order while the rank's next batch call runs.  Two gloo ranks, six steps with lists
    of different lengths, the rank's buffer overwritten right after each submit: both ranks end with the LAST step's results
    of all frames, in global order."""
    import torch.multiprocessing as mp

    sys.path.insert(0, ROOT)
    from librectify_amd import LINE_DTYPE

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 17
    n_steps = 6
    procs = [ctx.Process(target=_worker_async, args=(r, 2, port, n_steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    step = n_steps - 1
    exp = []
    for i in range(5):
        n = (i + step) % 7
        a = np.zeros(n, LINE_DTYPE)
        a["x1"] = np.arange(n, dtype=np.float32) + 100 * i + step
        a["group_id"] = i
        exp.append(a.tobytes())
    for rank, lines, tf in res:
        assert lines == exp, rank
        assert tf == np.full((5, 6, 3), float(step), np.float32).tobytes()
