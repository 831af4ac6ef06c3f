#!/bin/bash
# parity of the concurrent multi-source launch, then timelines of single 4K frames at several thresholds
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_source or flood or storage_tier or natural or full_size or carries or outgrows or batch" > gpurun_out/r04_multi_tests2.log 2>&1; rc=$?
tail -5 gpurun_out/r04_multi_tests2.log
[ $rc -ne 0 ] && exit $rc
for cfg in off 40 60 80 100; do
  for seed in 1 4; do
    if [ $cfg = off ]; then export LIBRECTIFY_FLOOD_MULTI=0; unset LIBRECTIFY_FLOOD_MULTI_MIN; else export LIBRECTIFY_FLOOD_MULTI=1; export LIBRECTIFY_FLOOD_MULTI_MIN=$cfg; fi
    bash tools/single_frame_trace.sh gpurun_out/sft_m${cfg}_s$seed $seed > /dev/null 2>&1
    echo "==== multi $cfg seed $seed: $(tail -1 gpurun_out/sft_m${cfg}_s$seed/run.txt | cut -c1-60) $(tail -1 gpurun_out/sft_m${cfg}_s$seed/run.txt | sed 's/.*multi_source_walks/multi/')"
  done
done
