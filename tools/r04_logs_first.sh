#!/bin/bash
# re-walks from the log: parity first (label tests with the logs on), then single 4K frames off / on
set -o pipefail
mkdir -p gpurun_out
LIBRECTIFY_FLOOD_LOGS=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -q --deselect tests/test_gpu_parity.py::test_multi_source_rewalks_change_the_time_not_the_labels > gpurun_out/logs_parity.txt 2>&1 || { tail -40 gpurun_out/logs_parity.txt; exit 1; }
tail -3 gpurun_out/logs_parity.txt
for m in 0 1 0 1; do
  echo "== LIBRECTIFY_FLOOD_LOGS=$m"
  LIBRECTIFY_FLOOD_LOGS=$m timeout -k 10 300 python tools/run4k_seeds.py 2>&1 | awk '{print $1, $2, $3, $4, $5, $6, $7, $NF, $(NF-1), $(NF-2), $(NF-3), $(NF-4), $(NF-5)}' || exit 1
done 2>&1 | tee gpurun_out/logs_ab.txt
