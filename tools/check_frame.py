"""Named frames (tools/frames.py) against the oracle: label image after the staged flood, records after the fit, the full call;
counters and times of each.  usage: check_frame.py <name> [<name> ...]   (LIBRECTIFY_* knobs apply)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import frames
import oracle_lib as O
import librectify_amd as L

ctx = L.Context(0)
ctx.set_seed(0)
ctx.set_stage_timing(True)
bad = 0
for name in sys.argv[1:]:
    img, min_len = frames.make(name)
    t = time.perf_counter()
    ref = O.find_line_segments(img, num_threads=8)
    cpu_ms = (time.perf_counter() - t) * 1e3
    for rep in range(2):
        ctx.stage_filter_host(img)
        ctx.stage_seeds()
        ctx.stage_flood()
        lab = ctx.download(L.BUF_LABEL)
        c = ctx.stage_counters()
        lines = ctx.stage_fit()
        ok_l = bool((lab == ref["label"]).all())
        ok_r = lines.tobytes() == ref["lines"].tobytes()
        t = time.perf_counter()
        got = ctx.find_line_segment_groups(img, min_len)
        wall = (time.perf_counter() - t) * 1e3
        st = ctx.stage_times()
        print("%-14s rep %d labels %s records %s | oracle detect %.0f ms (8 threads) | call %.2f ms flood %.2f fit %.2f | rounds %d giant steps %d held %d slabs %d tail %d tier2 %d"
              % (name, rep, "ok" if ok_l else "DIFFER (%d px)" % int((lab != ref["label"]).sum()), "ok" if ok_r else "DIFFER", cpu_ms, wall,
                 float(st[L.T_FLOOD]), float(st[L.T_FIT]), c["flood_rounds"], c["giant_steps"], c["giants_held"], c["slabs"], c["ordered_tail_seeds"],
                 c["second_tier_seeds"]), flush=True)
        bad += (not ok_l) + (not ok_r)
print("FAILED" if bad else "all equal")
sys.exit(1 if bad else 0)
