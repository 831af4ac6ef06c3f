#!/bin/bash
# timeline of a single 4K frame (seed $1, default 1) with the logs off and on
mkdir -p gpurun_out
S=${1:-1}
for m in 0 1; do
  export LIBRECTIFY_FLOOD_LOGS=$m
  echo "== LIBRECTIFY_FLOOD_LOGS=$m" 
  bash tools/single_frame_trace.sh gpurun_out/sft_logs${m}_s$S $S | grep -E "flood_|rewalk" > gpurun_out/sft_logs${m}_s$S.txt
  tail -1 gpurun_out/sft_logs${m}_s$S/run.txt
done
