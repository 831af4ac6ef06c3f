#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for g in 512 32 0; do
  O=gpurun_out/tg_$g; rm -rf $O; mkdir -p $O
  LIBRECTIFY_FLOOD_TEAM_GRID=$g rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/bench.json 2>/dev/null
  f=$(find $O/p -name "*kernel_stats.csv" | head -1)
  python3 - $f $g $O/bench.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== TEAM_GRID=%s: %.0f Mpix/s under the profiler, kernel-time sum %.0f ms" % (sys.argv[2], json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])["value"], tot / 1e6))
for r in rows:
    if any(k in r["Name"] for k in ("flood_explore_team", "flood_commit_pixels", "flood_survivors", "flood_explore_kernel")):
        print("    %-40s calls %6s avg %7.1f us  %5.1f %%" % (r["Name"].split("::")[-1][:40], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
  rm -rf $O/p
done
