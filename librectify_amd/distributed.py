"""Multi-GPU plumbing: frames are independent, so ranks shard them and exchange results once.

SURVEY.md §8e: contiguous blocks of ceil(B/G) frames per rank, no data-path collective; one final
gather of (n_lines, LineSegment[n], ImageTransform) per frame over torch.distributed (backend
"nccl" = RCCL over xGMI on GPUs, "gloo" on CPU for tests).
"""
import numpy as np

from . import LINE_DTYPE


def shard_range(n_items, rank, world):
    """[begin, end) of the contiguous block of rank `rank`."""
    per = (n_items + world - 1) // world
    b = min(n_items, rank * per)
    e = min(n_items, b + per)
    return b, e


def gather_results(lines_per_frame, transforms, n_total, device=None):
    """lines_per_frame: list of LINE_DTYPE arrays for this rank's frames (in shard order);
    transforms: float32 array [n_local, 6, 3].  Returns (list of LINE_DTYPE arrays for all n_total
    frames in global order, float32 [n_total, 6, 3]) on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    n_local = len(lines_per_frame)
    dev = device if device is not None else torch.device("cpu")
    # two collectives: the counts and transforms first (19 floats a frame) -- every rank then knows the longest list of
    # all and sizes the payload by it -- then the segments.  (Round 4 asked the ranks for that maximum with an all_reduce
    # of its own: three collectives and three host round trips a step cost the one-rank rehearsal 3-6 %.)
    meta = np.zeros((per, 1 + 18), np.float32)
    meta[:n_local, 0] = [len(l) for l in lines_per_frame]
    if n_local:
        meta[:n_local, 1:] = np.asarray(transforms, np.float32).reshape(n_local, 18)
    t_meta = torch.from_numpy(meta).to(dev)
    g_meta = torch.empty((world,) + tuple(t_meta.shape), dtype=t_meta.dtype, device=dev)
    dist.all_gather_into_tensor(g_meta.view(world * per, 19), t_meta)
    m_all = g_meta.cpu().numpy()
    cap = max(1, int(m_all[:, :, 0].max()))
    payload = np.zeros((per, cap), LINE_DTYPE)
    for i, l in enumerate(lines_per_frame):
        payload[i, : len(l)] = l
    t_pay = torch.from_numpy(payload.view(np.uint8).reshape(per, cap * LINE_DTYPE.itemsize)).to(dev)
    g_pay = torch.empty((world * per, cap * LINE_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(g_pay, t_pay)
    p_all = g_pay.cpu().numpy().reshape(world, per, cap * LINE_DTYPE.itemsize)
    out_lines, out_tf = [], np.zeros((n_total, 6, 3), np.float32)
    for r in range(world):
        b, e = shard_range(n_total, r, world)
        for i in range(e - b):
            n = int(m_all[r, i, 0])
            out_lines.append(p_all[r, i].view(LINE_DTYPE)[:n].copy())
            out_tf[b + i] = m_all[r, i, 1:].reshape(6, 3)
    return out_lines, out_tf


class GatherWorker:
    """The path's one exchange step off the critical path: a thread of its own takes the steps' results in order and gathers
    them over the process group (gather_results) while the rank goes on with its next batch of frames.  A rank issues its
    collectives from this one thread, in step order -- the same order on every rank; `drain()` waits for everything submitted
    (call it on every rank before any collective of the main thread, e.g. a barrier) and re-raises what a gather raised;
    `results` keeps the last gather's return value when `keep` is set."""

    def __init__(self, device=None, keep=False):
        import queue
        import threading

        self.device = device
        self.keep = keep
        self.results = None
        self.err = None
        self.q = queue.Queue()
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        import torch

        if self.device is not None and torch.device(self.device).type == "cuda":
            torch.cuda.set_device(self.device)
        while True:
            job = self.q.get()
            try:
                if job is None:
                    return
                if self.err is None:
                    r = gather_results(job[0], job[1], job[2], device=self.device)
                    if self.keep:
                        self.results = r
            except Exception as e:  # (reported by drain() on the rank's main thread)
                self.err = e
            finally:
                self.q.task_done()

    def submit(self, lines_per_frame, transforms, n_total):
        """the arrays must stay untouched until the gather has run: hand over copies of buffers that are reused"""
        self.q.put((lines_per_frame, transforms, n_total))

    def drain(self):
        self.q.join()
        if self.err is not None:
            raise self.err

    def close(self):
        self.drain()
        self.q.put(None)
        self.t.join()
