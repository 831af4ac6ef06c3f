#!/bin/bash
# usage: tools/exp_bench.sh <out> [bench args ...] -- "VAR=VAL;VAR=VAL" ...   the headline leg (no extra legs) per setting, twice
out=$1; shift
args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
mkdir -p gpurun_out; : > gpurun_out/$out
for setting in "base" "$@" "base"; do
  if [ "$setting" = "base" ]; then envs=(); else IFS=';' read -ra envs <<< "$setting"; fi
  for rep in 1 2; do
    v=$(env "${envs[@]}" python3 bench.py --no-cpu-baseline --no-extra-legs "${args[@]}" 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['stage_ms_per_frame'])")
    echo "$setting | $v" >> gpurun_out/$out
  done
done
cat gpurun_out/$out
