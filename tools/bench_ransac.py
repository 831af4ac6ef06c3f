import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import bench, librectify_amd as L
ctx = L.Context(0)
for k, v in bench.ransac_rates(ctx).items():
    print(k, v, "frac of fp32 peak %.3f" % (v["line_evaluations_per_s"] * 15 / 157.3e12))
for k, v in bench.cht_rates(ctx).items():
    print("cht", k, v)
