"""One named frame (tools/frames.py) through the frame call: wall time of the repetitions, stage times, counters.
LIBRECTIFY_FLOOD_DEBUG=1 prints the rounds.  usage: run_frame.py <name> [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import frames
import librectify_amd as L

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
img, min_len = frames.make(name)
ctx = L.Context(0)
ctx.set_stage_timing(True)
dts = []
for rep in range(reps):
    t = time.perf_counter()
    got = ctx.find_line_segment_groups(img, min_len)
    dts.append((time.perf_counter() - t) * 1e3)
c = ctx.stage_counters()
print("%-14s wall %s ms  flood %.2f ms  lines %d" % (name, " ".join("%.2f" % d for d in dts), float(ctx.stage_times()[L.T_FLOOD]), len(got)))
print("   ", c)
print("   ", ctx.stage_times().round(3))
