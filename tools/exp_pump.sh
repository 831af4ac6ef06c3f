#!/bin/bash
# the resident-frames batch beside a DMA pump: tools/exp_pump.sh <out> pump ...   ("none" = no pump)
out=gpurun_out/$1; shift; : > $out
for p in "$@"; do
  for rep in 1 2; do
    if [ "$p" = "none" ]; then a=(); else a=(--dma-pump "$p"); fi
    r=$(python3 bench.py --no-cpu-baseline --no-extra-legs --steps 10 --warmup 3 --host-memory device "${a[@]}" 2>/tmp/pump_err | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
    echo "$p | $r | $(grep 'dma pump' /tmp/pump_err)" >> $out
  done
done
cat $out
