import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, frames, oracle_lib as O, librectify_amd as L
names = sys.argv[1:]
ctx = L.Context(0); ctx.set_seed(0)
for name in names:
    img, ml = frames.make(name)
    ref = O.find_line_segments(img, num_threads=8)
    for rep in range(3):
        ctx.stage_filter_host(img); ctx.stage_seeds(); ctx.stage_flood()
        lab = ctx.download(L.BUF_LABEL)
        c = ctx.stage_counters()
        lines = ctx.stage_fit()
        ok = lines.tobytes() == ref["lines"].tobytes()
        print(name, rep, "labels", bool((lab == ref["label"]).all()), "records", ok, len(lines), len(ref["lines"]), c["flood_rounds"], c["giant_steps"], c["giants_held"], flush=True)
        if not ok and len(lines) == len(ref["lines"]):
            for i in range(len(lines)):
                if lines[i].tobytes() != ref["lines"][i].tobytes():
                    print("   comp", i, "got", lines[i], "want", ref["lines"][i])
