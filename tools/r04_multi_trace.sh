#!/bin/bash
# timelines of single 4K frames (bench seeds 1 and 4) with and without multi-source re-walks, at several way-point thresholds
mkdir -p gpurun_out
for cfg in off 60 100 140; do
  for seed in 1 4; do
    if [ $cfg = off ]; then export LIBRECTIFY_FLOOD_MULTI=0; unset LIBRECTIFY_FLOOD_MULTI_MIN; else export LIBRECTIFY_FLOOD_MULTI=1; export LIBRECTIFY_FLOOD_MULTI_MIN=$cfg; fi
    echo "==== multi $cfg seed $seed"
    bash tools/single_frame_trace.sh gpurun_out/sft_m${cfg}_s$seed $seed 2>&1 | grep "flood_\|copyBuffer" | awk '{print $1, $5, $7, $8, $10}' > gpurun_out/sft_m${cfg}_s$seed.txt
    cat gpurun_out/sft_m${cfg}_s$seed/run.txt | tail -1 | cut -c1-400
    awk '/flood_explore_kernel|flood_explore_team/ {printf "%s:%s ", substr($4,15,6), $3}' gpurun_out/sft_m${cfg}_s$seed.txt; echo
    awk 'BEGIN{a=0} /flood_/ {if (a==0) a=$1; b=$1+$3} END{print "flood span us", b-a}' gpurun_out/sft_m${cfg}_s$seed.txt
  done
done
