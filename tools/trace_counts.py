"""Per-kernel launch counts and time of the LAST frame-sized slice of a rocprofv3 kernel trace (tools/run4k.py runs 3 frames)."""
import csv, sys, glob, collections, re
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# frames are delimited by the filter kernel
idx = [i for i, r in enumerate(rows) if "filter_lanes_kernel" in r["Kernel_Name"]]
last = rows[idx[-1]:]
by = collections.OrderedDict()
for r in last:
    m = re.search(r"(\w+_kernel)\b", r["Kernel_Name"])
    n = m.group(1) if m else r["Kernel_Name"].split("(")[0][-48:]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    by.setdefault(n, [0, 0])
    by[n][0] += 1
    by[n][1] += d
print("ops in last frame:", len(last), " span %.3f ms" % ((int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6),
      " summed %.3f ms" % (sum(v[1] for v in by.values()) / 1e6))
for n, (c, d) in by.items():
    print("  %-50s n=%3d  %8.1f us" % (n, c, d / 1e3))

if len(sys.argv) > 3:  # timeline with gaps
    prev = None
    for r in last:
        m = re.search(r"(\w+_kernel)\b", r["Kernel_Name"])
        n = m.group(1) if m else r["Kernel_Name"].split("(")[0][-30:]
        s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s0 - prev) / 1e3 if prev else 0.0
        print("   +%7.1f us gap | %8.1f us  %s" % (gap, (e0 - s0) / 1e3, n))
        prev = e0
