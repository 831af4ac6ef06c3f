#!/usr/bin/env python3
"""bench.py — librectify hot path on MI355X: Mpix/s end-to-end (detect + VP) on 4K frames.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  One step = one pass of the hot path
(find_line_segment_groups + compute_rectification_transform) over this rank's batch
of `--frames` distinct synthetic 3840x2160 frames.

`value` is measured FROM HOST POINTERS: the timed call is lr_find_line_segment_groups_batch_host on
frames in ordinary (pageable) host memory, so the upload of every frame (pinned staging + H2D) and the
download of its results are inside the timed region, as in the reference's only entry point
(interface.cpp:46, image.cpp:11-19).  The same workload on page-locked host frames and on frames already
resident in HBM is measured in short extra legs and reported under other keys, never as `value`.
Frames are independent, so ranks shard them with no data-path collective (weak scaling: per-GPU work is
fixed); the only exchange is the final gather of the per-frame results over RCCL.

Rank 0 prints ONE JSON line carrying, besides the contract's keys:
  roofline     — the fused filter kernel (HBM-bound): algorithmic 18 B/px over the
                 kernel's mean duration, measured with HIP events on the library's own stream; and the
                 same with the bytes the kernel really moves (PMC pass under profiles/);
  cpu_baseline — the CPU oracle (a restatement of the reference; the Eigen reference itself cannot be
                 built here) timed on a bounded sample of the same workload, threaded and serial;
  ransac_hypotheses_per_s — BASELINE metric (ii) at N = 1 000 / 10 000 and N = 20 000 / 100 000;
  config4_batch1080 — BASELINE configs[3]: 512 frames 1920x1080 sharded over the ranks (strong scaling),
                 host pointers, one gather.

  single_frame — ONE frame through the reference's own six-symbol entry (find_line_segment_groups +
                 compute_rectification_transform): wall ms and Mpix/s from a pageable and from a
                 page-locked buffer, mean over the four bench frames;
  flood        — the ordered flood of those frames: ms, component pixels per second, pixels walked per pixel labelled;
  roofline_ransac — 15 flop per (hypothesis, line) x line evaluations per second against the fp32 vector peak;
  roofline_8k  — the filter kernel on 8192x8192 frames (1.2 GB of algorithmic traffic per launch: nothing of it
                 fits the 256 MiB Infinity Cache);
  cht          — the diamond-space estimator (lr_estimate_line_pencils_cht): ms per grouping call and votes per second.

`--config batch1080` makes that last workload the timed one (512 frames over N ranks, "scaling": "strong").

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N ranks itself (`python -m
torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a child process, before anything here has
touched the GPU) and relays rank 0's JSON line.  WORLD_SIZE set but different from --gpus is refused.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# HIP maps its streams onto this many hardware queues (default 4); kernels of streams that share a queue run one after
# the other.  The batch entry keeps one stream per frame in flight, so give them a queue each (read at HIP start-up).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W4K, H4K = 3840, 2160
ALGO_BYTES_PER_PX = 18.0  # 4 read + 12 (dx,dy,mag) + 1 (bin) + 1 (peak candidate)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PMC_FILE = os.path.join("profiles", "r05_pmc_filter_traffic.txt")
PMC_FILE_8K = os.path.join("profiles", "r05_pmc_filter_traffic_8k.txt")
ROCPROF_LEG = os.path.join("profiles", "r05_kernel_stats_roofline_leg.csv")  # rocprofv3 --kernel-trace --stats of `bench.py --roofline-only`


def make_frames(n, w, h, seed0, bases=4, out=None):
    """n frames in distinct buffers: `bases` generated ones (seeds seed0, seed0 + 1, ...) and their three flips, repeated
    as often as needed (generating a 4K frame takes seconds; what matters is that every frame is its own buffer).
    Written into `out` ([n, h, w] float32) if given."""
    from librectify_amd import synth

    base = [synth.frame(w, h, seed0 + i) for i in range(min(n, bases))]
    res = out if out is not None else np.empty((n, h, w), np.float32)
    for i in range(n):
        b = base[i % len(base)]
        k = (i // len(base)) % 4
        res[i] = b if k == 0 else (b[:, ::-1] if k == 1 else (b[::-1, :] if k == 2 else b[::-1, ::-1]))
    return res


def pmc_traffic(w, h):
    """HBM bytes per filter launch from the committed PMC passes (4K and 8192 x 8192 frames); None for other sizes."""
    path = os.path.join(ROOT, PMC_FILE if (w, h) == (W4K, H4K) else PMC_FILE_8K)
    if (w, h) not in ((W4K, H4K), (8192, 8192)) or not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith("traffic_bytes_per_launch"):
            return float(line.split()[1])
    return None


def rocprof_filter_ms(w, h):
    """Average duration of the filter kernel in the committed rocprofv3 summary of the roofline leg (4K only)."""
    path = os.path.join(ROOT, ROCPROF_LEG)
    if (w, h) != (W4K, H4K) or not os.path.exists(path):
        return None
    import csv

    for row in csv.DictReader(open(path)):
        if "filter_lanes_kernel" in row["Name"]:
            return float(row["AverageNs"]) * 1e-6
    return None


_CPU_LEG = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import oracle_lib as O
frames = np.load(%(frames)r)
h, w = frames.shape[1:]
names = ["gradients", "directions", "seeds", "components", "fitting", "ransac", "total"]
def leg(threads, budget, max_frames):
    O.find_line_segment_groups(frames[0], %(min_length)r, num_threads=threads, seed=0)  # first touch of the work planes: not timed
    done, t0, stage = 0, time.perf_counter(), np.zeros(7)
    while True:
        img = frames[done %% len(frames)]
        lines, times = O.find_line_segment_groups(img, %(min_length)r, num_threads=threads, seed=0)
        O.compute_rectification_transform(lines, w, h)
        stage += times
        done += 1
        if time.perf_counter() - t0 >= budget or done >= max_frames:
            break
    el = time.perf_counter() - t0
    return {"threads": threads, "frames": done, "seconds": round(el, 2), "Mpix_per_s": round(done * w * h / el / 1e6, 3),
            "stage_ms_per_frame": {k: round(float(v) / done, 2) for k, v in zip(names, stage)}}
cores = int(O.max_threads())
counts = sorted({c for c in (8, 32, cores) if c <= cores})
print(json.dumps({"cores": cores, "legs": [leg(c, %(budget)r, 24) for c in counts], "serial": leg(-1, %(budget)r, 6)}))
"""


def cpu_baseline(frames, w, h, min_length, budget_s=5.0):
    """CPU oracle on a bounded sample of the same workload (kind 'port': the Eigen reference is unbuildable here).  Only
    this leg of bench.py touches oracle/.  Run in a CHILD process, so that the oracle's OpenMP runtime starts with thread
    binding (OMP_PROC_BIND=close, OMP_PLACES=cores: the child inherits this rank's cores, i.e. one NUMA node) and shares
    nothing with this process's threads; the oracle keeps its frame-sized work planes from call to call, first touched by
    the threads that fill them.  Legs: 8, 32 and all threads -- `value` is the BEST of them, with its thread count -- and
    the reference's serial mode num_threads = -1."""
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "frames.npy")
        np.save(path, np.ascontiguousarray(frames))
        code = _CPU_LEG % {"root": ROOT, "frames": path, "min_length": float(min_length), "budget": float(budget_s)}
        env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        return {"error": p.stderr[-800:]}
    r = json.loads(p.stdout.strip().splitlines()[-1])
    best = max(r["legs"], key=lambda l: l["Mpix_per_s"])
    return {
        "value": best["Mpix_per_s"],
        "unit": "Mpix/s",
        "cores": int(best["threads"]),
        "kind": "port",
        "sample": "%d frame(s) %dx%d, oracle find_line_segment_groups+compute_rectification_transform, %d OpenMP threads bound to cores (best of %s), %.1f s"
                  % (best["frames"], w, h, best["threads"], [l["threads"] for l in r["legs"]], best["seconds"]),
        "stage_ms_per_frame": best["stage_ms_per_frame"],
        "by_threads": {str(l["threads"]): l["Mpix_per_s"] for l in r["legs"]},
        "host_cores_available": r["cores"],
        "serial": {
            "value": r["serial"]["Mpix_per_s"],
            "unit": "Mpix/s",
            "cores": 1,
            "sample": "%d frame(s), num_threads = -1 (the reference's serial mode, threading.h:24-27), %.1f s" % (r["serial"]["frames"], r["serial"]["seconds"]),
            "stage_ms_per_frame": r["serial"]["stage_ms_per_frame"],
        },
    }


def ransac_rates(ctx):
    """BASELINE metric (ii): hypotheses scored against all N lines per second, through lr_ransac_best (upload of the
    pencil table, scoring kernel, argmax, one synchronisation: what one peeling round costs)."""
    import librectify_amd as L
    from librectify_amd import synth

    out = {}
    for n, n_iter in [(1000, 10000), (20000, 100000)]:
        lines = np.ascontiguousarray(synth.random_segments(n, 42), L.LINE_DTYPE)
        xs = np.concatenate([lines["x1"], lines["x2"]])
        ys = np.concatenate([lines["y1"], lines["y2"]])
        cx, cy = xs.min() + 0.5 * (xs.max() - xs.min()), ys.min() + 0.5 * (ys.max() - ys.min())
        sc = max(xs.max() - xs.min(), ys.max() - ys.min())
        norm = lines.copy()
        for a, c0 in (("x1", cx), ("x2", cx), ("y1", cy), ("y2", cy)):
            norm[a] = (lines[a] - np.float32(c0)) / np.float32(sc)
        idx = np.arange(n, dtype=np.int32)
        tol = float(np.float32(1.0) - np.float32(np.cos(np.float32(2.0) / np.float32(180.0) * np.float32(np.pi))))
        ctx.ransac_best(norm, idx, tol, n_iter, 42)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.ransac_best(norm, idx, tol, n_iter, 42)
        dt = (time.perf_counter() - t0) / reps
        out["N%d_hyp%d" % (n, n_iter)] = {"hypotheses_per_s": round(n_iter / dt, 1), "line_evaluations_per_s": round(n_iter / dt * n, 1), "ms_per_solve": round(dt * 1e3, 4)}
    return out


FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak (no MFMA on this path)
RANSAC_FLOP_PER_EVAL = 15.0     # per (hypothesis, line)


def spawn_command(argv, n, port):
    """The child process that runs `n` ranks of this script on one node (one rank per GPU)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N > 1 without a launcher: start the N ranks as a child process and relay rank 0's line.  Nothing in this
    process has touched the GPU (and nothing will): no re-exec, the parent only waits."""
    import subprocess

    cmd = spawn_command(argv, args.gpus, free_port())
    if args.dry_run_spawn:
        print(json.dumps({"spawn": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in p.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            print(out, file=sys.stderr)
    rc = p.wait()
    if line is not None:
        print(line)
    elif rc == 0:
        rc = 1
    return rc


def cpu_list(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def numa_node_of(pci_bus_id):
    """NUMA node of a PCI device (sysfs), -1 if unknown."""
    try:
        return int(open("/sys/bus/pci/devices/%s/numa_node" % pci_bus_id.lower()).read())
    except (OSError, ValueError, AttributeError):
        return -1


def rank_cpus(local_rank, world, pci_bus_ids=None, available=None, nodes=None, numa="gpu"):
    """Host cores for this rank's lanes and staging threads: the cores of its GPU's NUMA node (sysfs), split evenly among
    the ranks whose GPUs sit on the same node (by their order among those GPUs, whatever the numbering of the devices);
    without that information, the rank's contiguous share of the cores this process may run on (all of them for a single
    rank).  numa = "other": the cores of the NEXT node instead (experiment: staging copies and the DMA that reads the
    staging buffers on different memory controllers); "none": no binding.
    pci_bus_ids: the bus id of every rank's device, by local rank (nodes: their NUMA nodes, for tests).  Returns
    (cores, how)."""
    avail = sorted(available if available is not None else os.sched_getaffinity(0))
    if not avail or numa == "none":
        return avail, "all"
    share = max(1, len(avail) // max(world, 1))
    fallback = (avail[local_rank * share : (local_rank + 1) * share] or avail) if world > 1 else avail
    if nodes is None and pci_bus_ids:
        nodes = [numa_node_of(b) for b in pci_bus_ids]
    if nodes and len(nodes) > local_rank and nodes[local_rank] >= 0:
        node = nodes[local_rank]
        use = node
        if numa == "other":
            try:
                n_nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
            except OSError:
                n_nodes = 1
            use = (node + max(1, n_nodes // 2)) % max(1, n_nodes)
        try:
            cpus = [c for c in cpu_list(open("/sys/devices/system/node/node%d/cpulist" % use).read()) if c in set(avail)]
        except (OSError, ValueError):
            cpus = []
        peers = [r for r in range(min(world, len(nodes))) if nodes[r] == node]
        k = peers.index(local_rank)
        sh = max(1, len(cpus) // len(peers))
        mine = cpus[k * sh : (k + 1) * sh]
        if mine:
            return mine, "numa node %d%s (%d rank(s) on it)" % (use, "" if use == node else " (the GPU sits on node %d)" % node, len(peers))
    return fallback, ("contiguous share" if world > 1 else "all")


def cht_rates(ctx):
    """The diamond-space estimator (lr_estimate_line_pencils_cht: votes of all lines into the LDS accumulators, four
    peeling rounds with argmax on the device and the removed lines' votes taken back): whole call."""
    import librectify_amd as L
    from librectify_amd import synth

    out = {}
    for n in (1000, 20000):
        lines = np.ascontiguousarray(synth.random_segments(n, 42), L.LINE_DTYPE)
        _, models, _, votes = ctx.estimate_line_pencils_cht(lines, d=128)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.estimate_line_pencils_cht(lines, d=128)
        dt = (time.perf_counter() - t0) / reps
        out["N%d_d128" % n] = {"ms_per_call": round(dt * 1e3, 4), "rounds": int(len(models)), "ms_per_solve": round(dt * 1e3 / max(1, len(models)), 4),
                               "votes_per_call": votes, "votes_per_s": round(votes / dt, 1)}
    return out


def single_frame_rates(L, ctx, frames, min_length):
    """One frame at a time through the reference's own entry points (the drop-in symbols find_line_segment_groups,
    compute_rectification_transform, release_line_segments): wall time from entry to return, pageable and page-locked
    source, mean over the bench frames (SURVEY.md §8d-i is defined on exactly this call pair)."""
    h, w = frames[0].shape
    out = {"frames": len(frames), "entry": "find_line_segment_groups + compute_rectification_transform (librectify.h), num_threads = 8"}
    pinned = ctx.host_alloc((len(frames), h, w))
    pinned[:] = frames
    for name, src in (("pageable", frames), ("page_locked", pinned)):
        per = []
        for f in src:
            for rep in range(4):
                t0 = time.perf_counter()
                lines = L.find_line_segment_groups(f, min_length, num_threads=8)
                L.compute_rectification_transform(lines, w, h)
                dt = time.perf_counter() - t0
                if rep > 0:
                    per.append(dt)
        ms = float(np.mean(per)) * 1e3
        out[name] = {"wall_ms": round(ms, 4), "Mpix_per_s": round(w * h / ms / 1e3, 2), "min_ms": round(float(np.min(per)) * 1e3, 4), "max_ms": round(float(np.max(per)) * 1e3, 4)}
    ctx.host_free(pinned)
    return out


def natural_frame_rates(L, ctx):
    """A NATURAL image at 4K next to the synthetic bench frames: the reference's own doc image (tests/golden/
    doc_image_gray.npy, the luma of doc/image.jpg) upsampled to 3840x2160 with a cubic spline.  Its floods are regions,
    not lines (walks of hundreds of tiles: second storage tier, hold-back of the weakest seeds), which is what the
    library's users feed it.  One frame at a time through the frame call, from a pageable buffer."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "doc_image_gray.npy")
    try:
        import scipy.ndimage as ndi
        g = np.load(path).astype(np.float32) / np.float32(256.0)
    except Exception as e:  # (scipy or the fixture missing: not a failure of the bench)
        return {"skipped": repr(e)}
    w, h = W4K, H4K
    img = np.ascontiguousarray(ndi.zoom(g, (h / g.shape[0], w / g.shape[1]), order=3).astype(np.float32)[:h, :w])
    ctx.set_stage_timing(True)
    wall, flood = [], []
    for rep in range(5):
        t0 = time.perf_counter()
        lines = ctx.find_line_segment_groups(img, float(max(w, h)) / 100.0)
        dt = time.perf_counter() - t0
        if rep > 1:
            wall.append(dt)
            flood.append(float(ctx.stage_times()[L.T_FLOOD]))
    c = ctx.stage_counters()
    ctx.set_stage_timing(False)
    return {"frame": "doc image upsampled to %dx%d (cubic spline)" % (w, h), "wall_ms": round(float(np.mean(wall)) * 1e3, 4),
            "Mpix_per_s": round(w * h / float(np.mean(wall)) / 1e6, 2), "flood_ms": round(float(np.mean(flood)), 4),
            "lines": int(len(lines)), "seeds": c["seeds"], "components": c["components"], "flood_rounds": c["flood_rounds"],
            "second_tier_walks": c["second_tier_seeds"], "walked_per_labelled": round(c.get("walked_px", 0) / max(1, c["labelled_px"]), 3)}


_CPU_FRAMES = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import oracle_lib as O
z = np.load(%(frames)r)
cores = int(O.max_threads())
counts = sorted({c for c in (8, 32, cores) if c <= cores})
out = {}
for name in z.files:
    img = z[name]
    h, w = img.shape
    best, best_t = None, 0
    for t in counts:
        for rep in range(2):  # (the first call touches the oracle's work planes)
            t0 = time.perf_counter()
            lines, _ = O.find_line_segment_groups(img, float(max(w, h)) / 100.0, num_threads=t, seed=0)
            O.compute_rectification_transform(lines, w, h)
            dt = (time.perf_counter() - t0) * 1e3
        if best is None or dt < best:
            best, best_t = dt, t
    out[name] = {"cpu_ms": round(best, 2), "threads": best_t}
print(json.dumps(out))
"""


def cpu_frame_ms(frames):
    """The CPU restatement (oracle/, the serial find_components of line_detector.cpp:92-122 inside it) on each of the given
    frames, in a child process with bound OpenMP threads like cpu_baseline: best of 8 / 32 / all threads, second call."""
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "frames.npz")
        np.savez(path, **frames)
        code = _CPU_FRAMES % {"root": ROOT, "frames": path}
        env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        return {"error": p.stderr[-800:]}
    return json.loads(p.stdout.strip().splitlines()[-1])


def worst_case_rates(L, ctx, with_cpu=True):
    """Frames the ordered flood likes least, one at a time through the frame call from a pageable buffer (the latency of a
    call depends on the content: INTEGRATION.md "Content-dependent latency"): 4K frames WITHOUT strong edges (soft blobs
    on a ramp: single floods of hundreds of thousands of pixels), a 4K frame of sixty bars 2000-3600 px long, a 4K frame
    that is nothing but a smooth ramp under blurred noise (every weak seed reaches regions of 100 000 pixels and more), and
    the two 1080p frames the latency fuzz of round 4 found slowest: soft blobs whose strong seeds own regions beyond the
    second storage tier, and a noiseless radial gradient (every pixel a seed of one magnitude, sixteen rings).  `cpu_ms` is
    the CPU restatement's time for the same frame and the same two calls (cpu_frame_ms), `cpu_over_gpu` their ratio."""
    from librectify_amd import synth

    yy, xx = np.mgrid[0:1080, 0:1920].astype(np.float64)
    radial = (1.0 - np.hypot(xx - 960, yy - 540) / np.hypot(960, 540)).astype(np.float32)
    frames = {"edgeless_4k": synth.region_frame(W4K, H4K, 504), "long_bars_4k": synth.long_bar_frame(W4K, H4K, 3), "ramp_4k": synth.ramp_frame(W4K, H4K),
              "regions_4k": synth.region_frame(W4K, H4K, 500), "regions_1080": synth.region_frame(1920, 1080, 500), "radial_gradient_1080": radial}
    out = {}
    ctx.set_stage_timing(True)
    prev_shape = None
    for name, img in frames.items():
        h, w = img.shape
        # (a context that has worked on 4K frames shrinks its workspace by itself at the eighth frame of a quarter the size --
        # 20 ms of reallocation that would land inside one timed call of the 1080p frames: shrink explicitly where the size changes)
        if prev_shape is not None and (h, w) != prev_shape:
            ctx.trim()
        prev_shape = (h, w)
        wall, flood, transform = [], [], []
        for rep in range(4):
            t0 = time.perf_counter()
            lines = ctx.find_line_segment_groups(img, float(max(w, h)) / 100.0)
            L.compute_rectification_transform(lines, w, h)
            dt = time.perf_counter() - t0
            if rep > 0:
                wall.append(dt)
                flood.append(float(ctx.stage_times()[L.T_FLOOD]))
        c = ctx.stage_counters()
        out[name] = {"frame": "%dx%d" % (w, h), "wall_ms": round(float(np.mean(wall)) * 1e3, 3), "max_ms": round(float(np.max(wall)) * 1e3, 3), "flood_ms": round(float(np.mean(flood)), 3),
                     "lines": int(len(lines)), "seeds": c["seeds"], "flood_rounds": c["flood_rounds"], "second_tier_walks": c["second_tier_seeds"],
                     "slabs": c["slabs"], "ordered_tail_seeds": c["ordered_tail_seeds"], "giants_held": c.get("giants_held", 0), "giant_steps": c.get("giant_steps", 0)}
    ctx.set_stage_timing(False)
    if with_cpu:
        cpu = cpu_frame_ms(frames)
        for name in frames:
            if name in cpu:
                out[name]["cpu_ms"] = cpu[name]["cpu_ms"]
                out[name]["cpu_threads"] = cpu[name]["threads"]
                out[name]["cpu_over_gpu"] = round(cpu[name]["cpu_ms"] / out[name]["wall_ms"], 2)
        if "error" in cpu:
            out["cpu_error"] = cpu["error"]
    out["note"] = ("find_line_segment_groups + compute_rectification_transform per call, mean of three calls on one context (a context hands what it "
                   "learned about a frame's floods to the next frame: the first call of its kind is slower -- DESIGN.md section 7)")
    return out


def flood_rates(L, ctx, frames):
    """The ordered flood (filter.cpp:110-153, line_detector.cpp:92-122) of the bench frames, one frame at a time through
    the frame call: device ms of the flood stage, labelled component pixels per second, and how many pixels the rounds
    walked for every pixel they labelled (re-walks of blocked seeds)."""
    ms, px, walked, rounds = [], [], [], []
    h, w = frames[0].shape
    ctx.set_stage_timing(True)
    for f in frames:
        for rep in range(3):
            ctx.find_line_segment_groups(f, float(max(w, h)) / 100.0)
            if rep > 0:
                ms.append(float(ctx.stage_times()[L.T_FLOOD]))
        c = ctx.stage_counters()
        px.append(c["labelled_px"])
        walked.append(c.get("walked_px", 0))
        rounds.append(c["flood_rounds"])
    ctx.set_stage_timing(False)
    m = float(np.mean(ms))
    return {"ms": round(m, 4), "component_pixels_per_s": round(float(np.mean(px)) / (m * 1e-3), 1), "labelled_px": float(np.mean(px)),
            "walked_px": float(np.mean(walked)), "walked_per_labelled": round(float(np.sum(walked)) / max(1.0, float(np.sum(px))), 3),
            "rounds": float(np.mean(rounds)), "frames": len(frames)}


def roofline_8k(L, ctx, torch, dev, base):
    """The filter kernel on 8192 x 8192 frames (the primary HBM evidence -- 1.2 GB of algorithmic
    traffic per launch, four times the Infinity Cache): three resident frames in turn, HIP events around each launch."""
    w = h = 8192
    if base is None:
        from librectify_amd import synth

        base = synth.frame(W4K, H4K, 1)
    big = np.tile(base, (4, 3))[:h, :w]  # (content does not matter to a streaming kernel; every buffer is distinct memory)
    d = torch.empty((3, h, w), dtype=torch.float32, device=dev)
    for i in range(3):
        d[i].copy_(torch.from_numpy(np.ascontiguousarray(np.roll(big, 17 * i, axis=1))))
    torch.cuda.synchronize()
    ms = []
    for lap in range(4):
        for b in range(3):
            ctx.stage_filter_device(d.data_ptr() + b * h * w * 4, w, h)
            ctx.synchronize()
            if lap > 0:
                ms.append(ctx.stage_times_partial())
    del d
    torch.cuda.empty_cache()
    k = float(np.mean(ms))
    ach = ALGO_BYTES_PER_PX * w * h / (k * 1e-3) / 1e9
    traffic = pmc_traffic(w, h)
    return {"bound": "hbm", "frame": "8192x8192", "kernel_ms": round(k, 5), "launches": len(ms), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PX * w * h,
            "traffic": traffic, "traffic_source": PMC_FILE_8K + " (as the 4K leg's: FETCH_SIZE x calibration + WRITE_SIZE, separate passes)",
            "achieved_moved_bytes": round(traffic / (k * 1e-3) / 1e9, 2) if traffic else None,
            "frac_moved_bytes": round(traffic / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["frames4k", "batch1080"], default="frames4k",
                    help="frames4k: BASELINE configs[2], --frames 4K frames per rank per step (weak scaling); "
                         "batch1080: configs[3], 512 frames 1920x1080 sharded over the ranks (strong scaling)")
    ap.add_argument("--frames", type=int, default=64, help="frames4k: frames (distinct buffers) per rank per step, i.e. per batch call: the call's "
                    "fill and drain (first uploads, last frames on fewer lanes) are inside the timed region")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--host-memory", choices=["pageable", "pinned", "device"], default="pageable",
                    help="where the timed region's frames live (the other kind and the device-resident rate are extra legs); "
                         "'device' = already in HBM: a diagnostic, NOT the metric (the line's workload says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="with a process group: leave the gather of the results out of the steps (what a process "
                    "group costs by being there, apart from what the gather costs)")
    ap.add_argument("--no-extra-legs", action="store_true")
    ap.add_argument("--sync-gather", action="store_true", help="with a process group: gather each step's results before the next step's call "
                    "(until round 5; default now: a worker thread gathers them while the next call runs, fence() waits for the last)")
    ap.add_argument("--dma-pump", default=None, metavar="DIR:MB",
                    help="experiment (profiles/r05_dma_interference.txt): beside the timed steps a thread keeps copying 32 MB pieces "
                         "h2d:MB = from page-locked host memory round a device region of MB megabytes, d2h:MB the other way, "
                         "d2d:MB inside HBM; its rate goes to stderr")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the roofline leg (the filter kernel alone): the command profiled for "
                         "profiles/*_kernel_stats_roofline_leg.csv, where rocprofv3's average must agree with kernel_ms")
    ap.add_argument("--flood-mode", type=int, default=None)
    ap.add_argument("--streams", type=int, default=5, help="frames in flight per GPU (one context + HIP stream + host thread each); "
                    "more than a handful only dilutes the 256 MB Infinity Cache that the flood's gathers live on "
                    "(round 4, with the lanes' rounds just in time: 4 / 5 / 6 / 7 lanes = 9.8 / 10.4 / 10.3 / 10.3 Gpix/s)")
    ap.add_argument("--staging-threads", type=int, default=12, help="host threads that stage pageable frames (num_threads of the batch call); "
                    "capped at this rank's share of the host cores")
    ap.add_argument("--dry-run-spawn", action="store_true", help="with --gpus N > 1 and no WORLD_SIZE: print the launch command instead of running it")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="with --gpus 1 and no WORLD_SIZE: run the ONE rank under torch.distributed.run all the same (started as a child before "
                         "anything here touches the GPU), so that the process group (RCCL) and the gather of the results run on the one GPU of a box")
    ap.add_argument("--numa", choices=["gpu", "other", "none"], default="gpu",
                    help="host cores of a rank (its lanes, uploader, staging threads; its frames are first touched there): the cores of "
                         "its GPU's NUMA node, of another node (experiment: copies and DMA on different memory controllers), or no binding")
    args = ap.parse_args(argv)

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus > 1 or args.rehearse_collective):
        return launch_ranks(args, argv)
    if env_world is not None and int(env_world) != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%s: refusing to run a different number of ranks than asked for" % (args.gpus, env_world), file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist

    import librectify_amd as L
    from librectify_amd import distributed as D

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # rehearsal hooks for a one-GPU box: LR_BENCH_BACKEND=gloo LR_BENCH_SINGLE_DEVICE=1 runs N ranks on device 0 with
    # the gather over gloo (RCCL refuses two ranks on one device); the driver's real runs use neither
    backend = os.environ.get("LR_BENCH_BACKEND", "nccl")
    if os.environ.get("LR_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)  # before the process group: RCCL binds each rank to its current device
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live
    # A process group exists whenever a launcher started this rank (WORLD_SIZE set, even to 1: `--rehearse-collective`);
    # the gather of the results runs exactly then.  A plain `python bench.py` (the driver's N = 1 run) has neither.
    pg = env_world is not None
    if pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    n_gpus = world if world > 1 else 1
    # host side of a rank: its lanes, its uploader and its staging threads stay on the cores of its GPU's NUMA node, and
    # no rank asks for more staging threads than its share of the cores (threads inherit the affinity set here; the frames
    # made below are first touched under it).  Since round 4 also for a single rank: on a two-socket host the staging
    # copies otherwise run wherever the scheduler puts them.
    bus_ids = []
    single = os.environ.get("LR_BENCH_SINGLE_DEVICE") == "1"
    for r in range(world):
        try:
            pr = torch.cuda.get_device_properties(0 if single else r)
            bus_ids.append("%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
        except Exception:
            bus_ids.append(None)
    cpus, cpus_how = rank_cpus(int(os.environ.get("LOCAL_RANK", "0")), world, bus_ids, numa=args.numa)
    if cpus and cpus_how != "all":
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            cpus_how += " (not applied)"
    args.staging_threads = max(1, min(args.staging_threads, len(cpus) if cpus else args.staging_threads))
    ranks_seen = world
    affinity = {"asked": len(cpus) if cpus else None, "before_first_collective": len(os.sched_getaffinity(0))}
    if pg:  # did the collective backend see every rank?
        t = torch.ones(1, dtype=torch.int32, device=cdev)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        ranks_seen = int(sum(int(x.item()) for x in got))
        # RCCL builds its communicator at the first collective and sets the calling thread's affinity while it does
        # ("Setting affinity for GPU 0 to ffffffff,..." in profiles/r04_rccl_world1.txt): the rank's own binding is applied
        # AGAIN behind it, and the line carries the mask before, behind RCCL, and as it was left
        affinity["after_first_collective"] = len(os.sched_getaffinity(0))
        if cpus and cpus_how != "all":
            try:
                os.sched_setaffinity(0, cpus)
            except OSError:
                pass
    affinity["in_timed_region"] = len(os.sched_getaffinity(0))
    gather_calls = [0]

    ctx = L.Context(local_rank)
    ctx.set_seed(0)
    S = max(1, args.streams)
    ctx.set_batch_streams(S)
    if args.flood_mode is not None:
        ctx.set_flood_mode(args.flood_mode)
    cfg = L.RectificationConfig()

    # (the path's one exchange step runs off the critical path: librectify_amd/distributed.py GatherWorker)
    gather_worker = D.GatherWorker(device=cdev) if (pg and not args.no_gather and not args.sync_gather) else None

    def fence():
        if gather_worker:
            gather_worker.drain()
        if pg:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    def timed(fn, steps):
        """barrier + sync, `steps` calls, barrier + sync; MAX over ranks"""
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        fence()
        el = time.perf_counter() - t0
        if pg:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    class Workload:
        """B frames of one size in pageable host memory, page-locked host memory and (4K only) HBM."""

        def __init__(self, w, h, B, seed0, bases, cap, want_device):
            self.w, self.h, self.B, self.cap = w, h, B, cap
            self.min_length = float(max(w, h)) / 100.0  # autorectify.cpp:134
            self.pageable = make_frames(B, w, h, seed0, bases) if B else np.empty((0, h, w), np.float32)
            self.pinned = None
            self.d = None
            self.want_device = want_device
            self.out = np.zeros((max(B, 1), cap), L.LINE_DTYPE)
            self.n_lines = np.zeros(max(B, 1), np.int32)
            self.tforms = np.zeros((max(B, 1), 6, 3), np.float32)

        def ensure_pinned(self):
            if self.pinned is None and self.B:
                self.pinned = ctx.host_alloc((self.B, self.h, self.w))
                self.pinned[:] = self.pageable

        def ensure_device(self):
            if self.d is None and self.B:
                self.d = torch.empty((self.B, self.h, self.w), dtype=torch.float32, device=dev)
                self.d.copy_(torch.from_numpy(self.pageable))
                torch.cuda.synchronize()

        def step(self, kind, n_total=None):
            """one pass over the B frames; returns nothing, results land in self.out / n_lines / tforms"""
            if self.B:
                if kind == "device":
                    _, n, tf = ctx.find_line_segment_groups_batch_device(self.d.data_ptr(), self.h * self.w, self.B, self.w, self.h, self.min_length, capacity=self.cap, cfg=cfg, out=self.out)
                else:
                    src = self.pageable if kind == "pageable" else self.pinned
                    _, n, tf = ctx.find_line_segment_groups_batch_host(src, self.min_length, num_threads=args.staging_threads, capacity=self.cap, cfg=cfg, out=self.out)
                self.n_lines[: self.B] = n
                for b in range(self.B):
                    self.tforms[b] = tf[b].as_array()
            if pg and not args.no_gather:  # the path's one exchange step: gather the per-frame results over the process group (RCCL on GPUs)
                gather_calls[0] += 1
                nt = n_total if n_total is not None else self.B * world
                if gather_worker:  # (the next step's call writes into self.out: the worker gets copies)
                    gather_worker.submit([self.out[b][: self.n_lines[b]].copy() for b in range(self.B)], self.tforms[: self.B].copy(), nt)
                else:
                    D.gather_results([self.out[b][: self.n_lines[b]] for b in range(self.B)], self.tforms[: self.B], nt, device=cdev)

    def make_batch1080():
        b, e = D.shard_range(512, rank, world)
        wl = Workload(1920, 1080, e - b, 1000 + b, 8, 2048, False)
        return wl, 512

    if args.config == "batch1080":
        wl, n_total = make_batch1080()
        scaling = "strong"
        total_frames = 512
    else:
        w = args.width or W4K
        h = args.height or H4K
        wl = Workload(w, h, args.frames, 1 + 100 * rank, 4, 8192, True)
        n_total = None
        scaling = "weak"
        total_frames = args.frames * n_gpus
    kind = args.host_memory
    if kind == "pinned":
        wl.ensure_pinned()
    if kind == "device":
        wl.ensure_device()

    if args.roofline_only:
        args.steps = args.warmup = 0
        args.no_cpu_baseline = args.no_extra_legs = True
    for _ in range(args.warmup):
        wl.step(kind, n_total)
    filt_ms = []
    stage_acc = np.zeros(L.T_COUNT)

    def timed_step():
        wl.step(kind, n_total)

    pump = None
    if args.dma_pump:
        import threading
        direction, mb = args.dma_pump.split(":")
        piece = 32 << 20
        n_pieces = max(1, (int(mb) << 20) // piece)
        host_t = torch.empty(piece, dtype=torch.uint8).pin_memory()
        dev_t = torch.empty((n_pieces, piece), dtype=torch.uint8, device=dev)
        dev_src = torch.empty(piece, dtype=torch.uint8, device=dev)
        pump = {"stop": False, "bytes": 0, "t": 0.0}

        def pump_body():
            st = torch.cuda.Stream(device=dev)
            i = 0
            t0 = time.perf_counter()
            with torch.cuda.stream(st):
                while not pump["stop"]:
                    for _ in range(4):
                        if direction == "h2d":
                            dev_t[i % n_pieces].copy_(host_t, non_blocking=True)
                        elif direction == "d2h":
                            host_t.copy_(dev_t[i % n_pieces], non_blocking=True)
                        else:
                            dev_t[i % n_pieces].copy_(dev_src, non_blocking=True)
                        i += 1
                    st.synchronize()
                    pump["bytes"] += 4 * piece
            pump["t"] = time.perf_counter() - t0

        pump["thread"] = threading.Thread(target=pump_body)
        pump["thread"].start()
    el = timed(timed_step, args.steps)
    if pump:
        pump["stop"] = True
        pump["thread"].join()
        sys.stderr.write("dma pump %s: %.1f GB/s beside the timed steps\n" % (args.dma_pump, pump["bytes"] / pump["t"] / 1e9))
    segs = float(np.mean(wl.n_lines[: max(wl.B, 1)]))
    # stage times of frames inside the pipeline: two extra steps with the stage timers on (they cost the frame seven event
    # records, some 40 us of idle GPU: not inside the timed region)
    if args.steps > 0 and not args.roofline_only:
        ctx.set_stage_timing(True)
        for _ in range(2):
            wl.step(kind, n_total)
            t = ctx.stage_times()  # lane 0's last frame of this step
            filt_ms.append(float(t[L.T_FILTER_KERNEL]))
            stage_acc[:] += t
        ctx.set_stage_timing(False)

    # ---- extra legs (outside the timed region; each bracketed like it) ------------------------------------------
    extra = {}
    if not args.no_extra_legs:
        legs = [k for k in ("pageable", "pinned", "device") if k != kind and (k != "device" or wl.want_device)]
        for k in legs:
            if k == "pinned":
                wl.ensure_pinned()
            if k == "device":
                wl.ensure_device()
            wl.step(k, n_total)
            steps = 3
            e = timed(lambda: wl.step(k, n_total), steps)
            extra[k] = round(total_frames * wl.w * wl.h * steps / e / 1e6, 3)
        if args.config == "frames4k":
            wl.d = None  # release the resident copy before the next workload
            if wl.pinned is not None:
                ctx.host_free(wl.pinned)
                wl.pinned = None
            torch.cuda.empty_cache()
            if world == 1 and args.frames < 256:
                # The same workload in calls of 256 frames: every batch call has a ramp (the first frames' uploads) and a
                # tail (the last frames on ever fewer lanes) of about one frame's latency, 6 ms of a 56 ms call of 64 frames.
                w3 = Workload(wl.w, wl.h, 256, 1 + 100 * rank, 4, 8192, False)
                w3.step("pageable")
                e = timed(lambda: w3.step("pageable"), 2)
                extra["pageable_256_per_call"] = round(256 * wl.w * wl.h * 2 / e / 1e6, 3)
                del w3
            w2, nt2 = make_batch1080()
            w2.step("pageable", nt2)
            e = timed(lambda: w2.step("pageable", nt2), 2)
            extra["config4_batch1080"] = {
                "workload": "512 frames 1920x1080 (seeds 1000+i), sharded over %d rank(s), pageable host pointers, one gather of the results" % n_gpus,
                "ms_per_pass": round(e / 2 * 1e3, 3),
                "frames_per_s": round(512 * 2 / e, 2),
                "Mpix_per_s": round(512 * 2 * 1920 * 1080 / e / 1e6, 3),
                "scaling": "strong",
                "segments_per_frame": float(np.mean(w2.n_lines[: max(w2.B, 1)])),
            }

    # roofline leg (after the timed region): the filter kernel alone, one launch at a time on one stream, over
    # distinct resident frames (more than the 256 MiB Infinity Cache holds), timed with the HIP events the library
    # records around the launch
    iso = []
    rw, rh = (W4K, H4K) if args.config == "batch1080" else (wl.w, wl.h)
    if rank == 0:
        nroof = 32
        wl.d = None
        torch.cuda.empty_cache()
        d_roof = torch.empty((nroof, rh, rw), dtype=torch.float32, device=dev)
        src = wl.pageable if (wl.w, wl.h) == (rw, rh) and wl.B >= 1 else make_frames(2, rw, rh, 1, 2)
        for i in range(nroof):
            d_roof[i].copy_(torch.from_numpy(src[i % len(src)]))
        torch.cuda.synchronize()
        # ... and four contexts in turn: a context writes its outputs (9 B/px: dx, dy, mask) into its own workspace, and one
        # 4K workspace's 75 MB would stay in the Infinity Cache from launch to launch; four are 300 MB
        roof_ctxs = [ctx] + [L.Context(local_rank) for _ in range(3)]
        for lap in range(3):
            for b in range(nroof):
                rc = roof_ctxs[(lap * nroof + b) % len(roof_ctxs)]
                rc.stage_filter_device(d_roof.data_ptr() + b * rh * rw * 4, rw, rh)
                rc.synchronize()
                if lap > 0:
                    iso.append(rc.stage_times_partial())
        for rc in roof_ctxs[1:]:
            rc.close()

    if rank == 0:
        w, h = wl.w, wl.h
        total_px = float(total_frames) * w * h * args.steps
        value = total_px / el / 1e6 if args.steps > 0 else None
        kdur_ms = float(np.mean(iso)) if iso else float("nan")
        achieved = ALGO_BYTES_PER_PX * rw * rh / (kdur_ms * 1e-3) / 1e9
        traffic = pmc_traffic(rw, rh)
        nfr = max(1, len(filt_ms))
        res = {
            "metric": "Mpix/s end-to-end (detect+VP) on 4K frames, batch call of %d frames, %d in flight" % (args.frames, max(1, args.streams)) if args.config == "frames4k" else "Mpix/s end-to-end (detect+VP), batch of 512 1920x1080 frames",
            "value": round(value, 3) if value is not None else None,
            "unit": "Mpix/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(el / max(1, args.steps) * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("%dx%d frames in %s HOST memory -> find_line_segment_groups + compute_rectification_transform per frame "
                             "(lr_find_line_segment_groups_batch_host): H2D of every frame and D2H of its results are inside the timed "
                             "region; default constants, refine=false, min_length=max(W,H)/100" % (w, h, kind)) if kind != "device" else
                            ("DIAGNOSTIC, not the metric: %dx%d frames already resident in HBM (no H2D in the timed region)" % (w, h)),
                "frames_per_step_all_gpus": total_frames,
                "frames_in_flight_per_gpu": S,
                "staging_threads": args.staging_threads,
                "ransac_iterations": 10000,
                "segments_per_frame": segs,
                "parallelism": {
                    "sharding": "frames in contiguous blocks over %d rank(s), one process per GPU, no data-path collective" % n_gpus,
                    "process_group": bool(pg),
                    "backend": (("nccl (RCCL)" if backend == "nccl" else backend) if pg else None),
                    "collective_tensors_on": (str(cdev) if pg else None),
                    "gather_ran": bool(pg and gather_calls[0] > 0),
                    "host_threads_affinity_cpus": affinity,
                    # what N ranks ask of the host's memory, the first shared resource of an 8-GPU node: a pageable frame is read
                    # by the staging copy, written into the staging buffer and read again by the DMA engine (3 x its bytes);
                    # a page-locked frame, or a pageable one that the batch call page-locks where it lies (the default since round 5;
                    # LIBRECTIFY_REGISTER_FRAMES=0 for the staging copy), once.  Two-socket DDR5-4800 x 24 channels is ~920 GB/s at best.
                    "host_dram_GBps_expected": None if (kind == "device" or args.steps == 0) else round((3.0 if kind == "pageable" and os.environ.get("LIBRECTIFY_REGISTER_FRAMES") == "0" else 1.0) * wl.B * w * h * 4.0 * args.steps / el / 1e9 * n_gpus, 1),
                    "gather": ("2 x all_gather of the per-frame results (counts and transforms, then the segments), inside the timed region, %d calls in this run%s" % (gather_calls[0], ", by a worker thread beside the next step's batch call (the timed region ends after the last gather)" if gather_worker else "")) if pg
                              else "none: a single rank started without a launcher has no process group and nothing to gather",
                },
            },
            "h2d": None if kind == "device" or args.steps == 0 else {
                "GBps_per_rank": round(wl.B * w * h * 4.0 * args.steps / el / 1e9, 3),
                "note": "frame bytes this rank sent up the link per second of the timed region (the link also carries nothing else of size: results are ~30 KB a frame); "
                        "PCIe Gen5 x16: 63 GB/s rated, 52.7 GB/s measured for back-to-back copies from page-locked memory",
                "host_memory": kind,
            },
            "other_rates_Mpix_per_s": {
                "note": "same workload, 3 steps each, outside the timed region: frames in the other kind of host memory, and frames already resident in HBM (no H2D: NOT the metric)",
                "host_pageable": extra.get("pageable"),
                "host_pinned": extra.get("pinned"),
                "device_resident": extra.get("device"),
                "host_pageable_256_frames_per_call": extra.get("pageable_256_per_call"),
            },
            "config4_batch1080": extra.get("config4_batch1080"),
            "roofline": {
                "bound": "hbm",
                "kernel": "filter_lanes_kernel (row-streaming fused 5x5 derivative + magnitude + bin + dilated mask + NMS candidates)",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x2.000 calibrated on a 2 GiB read of the same 4 B/lane shape)",
                "achieved_moved_bytes": round(traffic / (kdur_ms * 1e-3) / 1e9, 2) if traffic else None,
                "frac_moved_bytes": round(traffic / (kdur_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "kernel_ms": round(kdur_ms, 5),
                "rocprof": (lambda k: None if k is None else {
                    "source": ROCPROF_LEG, "kernel_ms": round(k, 5),
                    "frac": round(ALGO_BYTES_PER_PX * rw * rh / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_moved_bytes": round(traffic / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None})(rocprof_filter_ms(rw, rh)),
                "kernel_ms_in_pipeline": round(float(np.mean(filt_ms)), 5) if filt_ms else None,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PX * rw * rh,
                "frame": "%dx%d" % (rw, rh),
            },
            "stage_ms_per_frame": {
                k: round(float(stage_acc[i]) / nfr, 4)
                for k, i in [("filter", L.T_FILTER), ("seeds", L.T_SEEDS), ("flood", L.T_FLOOD), ("fit", L.T_FIT), ("ransac", L.T_RANSAC), ("total_device", L.T_TOTAL)]
            },
        }
        res["ranks_seen"] = ranks_seen
        res["host_cores_per_rank"] = {"count": len(cpus), "how": cpus_how, "staging_threads": args.staging_threads}
        if not args.no_extra_legs:
            rr = ransac_rates(ctx)
            res["ransac_hypotheses_per_s"] = rr
            res["roofline_ransac"] = {
                "bound": "fp32 vector ALU (no MFMA: no dense contraction on this path)",
                "flop_per_line_evaluation": RANSAC_FLOP_PER_EVAL,
                "peak": FP32_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                **{k: {"achieved": round(v["line_evaluations_per_s"] * RANSAC_FLOP_PER_EVAL / 1e12, 3),
                       "frac": round(v["line_evaluations_per_s"] * RANSAC_FLOP_PER_EVAL / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4)} for k, v in rr.items()},
                "note": "whole lr_ransac_best call: table upload, scoring launch, read-out of the best into page-locked memory, one wait",
            }
            res["cht"] = cht_rates(ctx)
            if args.config == "frames4k" and wl.B:
                res["single_frame"] = single_frame_rates(L, ctx, wl.pageable[: min(4, wl.B)], wl.min_length)
                res["flood"] = flood_rates(L, ctx, wl.pageable[: min(4, wl.B)])
                res["natural_frame"] = natural_frame_rates(L, ctx)
                res["worst_case"] = worst_case_rates(L, ctx, with_cpu=not args.no_cpu_baseline)
            res["roofline_8k"] = roofline_8k(L, ctx, torch, dev, wl.pageable[0] if wl.B and (wl.w, wl.h) == (W4K, H4K) else None)
        if not args.no_cpu_baseline and n_gpus == 1 and wl.B:
            res["cpu_baseline"] = cpu_baseline(wl.pageable[:2], w, h, wl.min_length)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if gather_worker:
        gather_worker.close()
    if pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
