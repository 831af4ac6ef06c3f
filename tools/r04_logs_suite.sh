#!/bin/bash
# re-walks from the log: the label tests with the logs on (union-find path, then the sweep path), then the frame timings
set -o pipefail
mkdir -p gpurun_out
D="not multi_source"
LIBRECTIFY_FLOOD_LOGS=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -q -k "$D" > gpurun_out/logs_parity.txt 2>&1 || { tail -40 gpurun_out/logs_parity.txt; exit 1; }
tail -2 gpurun_out/logs_parity.txt
LIBRECTIFY_FLOOD_LOGS=1 LIBRECTIFY_FLOOD_LOG_SWEEP=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "$D" > gpurun_out/logs_parity_sweep.txt 2>&1 || { tail -40 gpurun_out/logs_parity_sweep.txt; exit 1; }
tail -2 gpurun_out/logs_parity_sweep.txt
bash tools/r04_logs_frames.sh
