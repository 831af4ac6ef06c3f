#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 - <<'PY'
import os, sys; sys.path.insert(0, ".")
import librectify_amd as L
from librectify_amd import synth
ctx = L.Context(0)
v = []
for seed in range(1, 33):
    img = synth.frame(3840, 2160, seed)
    ctx.find_line_segment_groups(img, 38.4)
    c = ctx.stage_counters()
    v.append((c["second_tier_seeds"], c["flood_rounds"]))
print("second-tier walks (all rounds), rounds, bench frames 1..32 (logs on, single calls):", v)
ctx.set_flood_logs(0)
v = []
for seed in range(1, 33):
    img = synth.frame(3840, 2160, seed)
    ctx.find_line_segment_groups(img, 38.4)
    c = ctx.stage_counters()
    v.append((c["second_tier_seeds"], c["flood_rounds"]))
print("logs off (as in the lanes):", v)
PY
for g in 512 32 0; do
  O=gpurun_out/tg_$g; rm -rf $O; mkdir -p $O
  LIBRECTIFY_FLOOD_TEAM_GRID=$g rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/bench.json 2>/dev/null
  f=$(find $O/p -name "*kernel_stats.csv" | head -1)
  echo "== TEAM_GRID=$g: $(python3 -c "import json;print(json.loads(open('$O/bench.json').read().strip().splitlines()[-1])['value'])") Mpix/s under the profiler"
  grep -E "flood_explore_team|flood_commit_pixels|flood_survivors" $f | awk -F, '{print "   ", substr($1,1,60), "calls", $2, "avg us", $4/1000}'
  rm -rf $O/p
done
