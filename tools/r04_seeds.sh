#!/bin/bash
# the new seed order: parity at every size, then its launches' durations in single-frame traces (4K) and at 8192 x 8192
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "seed or full_path or golden or full_size or natural or baseline or capacity or random_small or laps" > gpurun_out/r04_seed_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r04_seed_tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/single_frame_trace.sh gpurun_out/sft_seeds 1 2>&1 | grep "seed_\|filter_lanes\|flood_init"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sft_8k/p -- python3 tools/run8k.py > gpurun_out/sft_8k_run.txt 2>/dev/null
find gpurun_out/sft_8k/p -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/sft_8k_kernel_stats.csv
rm -rf gpurun_out/sft_8k
grep -i "seed_\|rocprim" gpurun_out/sft_8k_kernel_stats.csv | cut -c1-60,150-260
grep "estimator 0" gpurun_out/sft_8k_run.txt | tail -1 | cut -c1-200
