"""Concurrency summary of a rocprofv3 kernel_trace.csv: wall span, union of kernel intervals, summed durations."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5  # ignore the first fraction (warm-up)
t0 = iv[0][0] + (iv[-1][1] - iv[0][0]) * skip
iv = [x for x in iv if x[0] >= t0]
span = max(e for _, e, _ in iv) - iv[0][0]
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e, _ in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
tot = sum(e - s for s, e, _ in iv)
print("kernels %d  span %.2f ms  union %.2f ms (%.0f%%)  summed %.2f ms  mean concurrency %.2f" % (
    len(iv), span / 1e6, union / 1e6, 100.0 * union / span, tot / 1e6, tot / union))
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in iv:
    by[n.split("(")[0][-40:]][0] += e - s
    by[n.split("(")[0][-40:]][1] += 1
for n, (d, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-42s n=%6d  total %8.2f ms  mean %8.1f us" % (n, c, d / 1e6, d / c / 1e3))
