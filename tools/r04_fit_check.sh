#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_tests_fit.txt 2>&1; rc=$?
tail -3 gpurun_out/r04_tests_fit.txt
[ $rc -ne 0 ] && exit $rc
bash tools/single_frame_trace.sh gpurun_out/sft_fit 1 | grep -E "fit_kernel|component_|filter_lines"
bash tools/trace_doc4k.sh gpurun_out/doc_fit 2>&1 | grep -E "fit_kernel|component_|filter_lines"
