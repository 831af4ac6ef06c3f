#!/bin/bash
# usage: tools/exp_env.sh <out> <frames...> -- VAR=VAL[,VAR=VAL] ...   one line per (setting, frame): wall of the last repetitions
out=$1; shift
frames=(); while [ "$1" != "--" ]; do frames+=("$1"); shift; done; shift
mkdir -p gpurun_out; : > gpurun_out/$out
for setting in "base" "$@"; do
  for f in "${frames[@]}"; do
    if [ "$setting" = "base" ]; then envs=(); else IFS=';' read -ra envs <<< "$setting"; fi
    line=$(env "${envs[@]}" python3 tools/run_frame.py $f 4 2>/dev/null | head -2 | tr '\n' ' ')
    echo "$setting | $line" | sed -e "s/'labelled_px'.*'walked_px'/'walked_px'/" -e "s/'multi_source.*'giants_held'/'giants_held'/" >> gpurun_out/$out
  done
done
