#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests_logs.txt 2>&1; rc=$?
tail -15 gpurun_out/r04_gpu_tests_logs.txt
exit $rc
