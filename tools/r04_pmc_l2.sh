#!/bin/bash
# L2 hits / misses of the frame's kernels inside the six-lane batch: frames resident in HBM vs frames arriving from page-locked host memory
export TMPDIR=/tmp
for kind in device pinned; do
  rm -rf gpurun_out/pmc_l2_$kind
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_l2_$kind -- python3 bench.py --steps 3 --warmup 1 --no-extra-legs --no-cpu-baseline --host-memory $kind > gpurun_out/pmc_l2_$kind.json 2> gpurun_out/pmc_l2_$kind.err
  python3 - gpurun_out/pmc_l2_$kind $kind <<'PY'
import csv, glob, sys, collections
d, kind = sys.argv[1:3]
fs = glob.glob(d + "/*/*counter_collection.csv")
if not fs:
    print(kind, "no counter file"); sys.exit(0)
hit = collections.defaultdict(float); miss = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    if r["Counter_Name"] == "TCC_HIT_sum": hit[k] += float(r["Counter_Value"]); n[k] += 1
    if r["Counter_Name"] == "TCC_MISS_sum": miss[k] += float(r["Counter_Value"])
tot_h, tot_m = sum(hit.values()), sum(miss.values())
print("%s: all kernels: L2 hits %.3e misses %.3e hit rate %.4f" % (kind, tot_h, tot_m, tot_h / max(1.0, tot_h + tot_m)))
for k in sorted(hit, key=lambda k: -(hit[k] + miss[k]))[:6]:
    print("   %-28s calls %5d hit rate %.4f  misses per call %.0f" % (k, n[k], hit[k] / max(1.0, hit[k] + miss[k]), miss[k] / max(1, n[k])))
PY
  rm -rf gpurun_out/pmc_l2_$kind
done
