"""Share of the wall span covered by flood_explore kernels (union), by round position, from a rocprofv3 kernel trace."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0 = iv[0][0] + (iv[-1][1] - iv[0][0]) * 0.5
iv = [x for x in iv if x[0] >= t0]
span = max(e for _, e, _ in iv) - iv[0][0]
def union(xs):
    xs = sorted(xs)
    if not xs: return 0
    u, cs, ce = 0, xs[0][0], xs[0][1]
    for s, e in xs[1:]:
        if s > ce:
            u += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    return u + ce - cs
ex = [(s, e) for s, e, n in iv if "flood_explore" in n]
big = [(s, e) for s, e in ex if e - s > 300e3]
print("span %.2f ms; explore kernels %d, union %.2f ms (%.0f%%), summed %.2f ms; those >300us: %d, union %.2f ms (%.0f%%), mean %.0f us" % (
    span / 1e6, len(ex), union(ex) / 1e6, 100 * union(ex) / span, sum(e - s for s, e in ex) / 1e6,
    len(big), union(big) / 1e6, 100 * union(big) / span, (sum(e - s for s, e in big) / max(1, len(big))) / 1e3))
oth = [(s, e) for s, e, n in iv if "flood_explore" not in n]
print("other kernels %d, union %.2f ms (%.0f%%), summed %.2f ms" % (len(oth), union(oth) / 1e6, 100 * union(oth) / span, sum(e - s for s, e in oth) / 1e6))
