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
    counts = np.zeros(per, np.int32)
    counts[:n_local] = [len(l) for l in lines_per_frame]
    cap_t = torch.tensor([int(counts.max()) if n_local else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
    cap = max(1, int(cap_t.item()))
    payload = np.zeros((per, cap), LINE_DTYPE)
    for i, l in enumerate(lines_per_frame):
        payload[i, : len(l)] = l
    meta = np.zeros((per, 1 + 18), np.float32)
    meta[:, 0] = counts
    if n_local:
        meta[:n_local, 1:] = np.asarray(transforms, np.float32).reshape(n_local, 18)
    t_pay = torch.from_numpy(payload.view(np.uint8).reshape(per, cap * LINE_DTYPE.itemsize)).to(dev)
    t_meta = torch.from_numpy(meta).to(dev)
    g_pay = [torch.empty_like(t_pay) for _ in range(world)]
    g_meta = [torch.empty_like(t_meta) for _ in range(world)]
    dist.all_gather(g_pay, t_pay)
    dist.all_gather(g_meta, t_meta)
    out_lines, out_tf = [], np.zeros((n_total, 6, 3), np.float32)
    for r in range(world):
        b, e = shard_range(n_total, r, world)
        m = g_meta[r].cpu().numpy()
        p = g_pay[r].cpu().numpy().reshape(per, cap * LINE_DTYPE.itemsize)
        for i in range(e - b):
            n = int(m[i, 0])
            out_lines.append(p[i].view(LINE_DTYPE)[:n].copy())
            out_tf[b + i] = m[i, 1:].reshape(6, 3)
    return out_lines, out_tf
