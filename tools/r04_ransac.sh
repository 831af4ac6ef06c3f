#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "ransac or grouping or full_path or golden or estimator or prosac or direct or cht or batch or baseline or laps or size_independent" > gpurun_out/r04_ransac_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r04_ransac_tests.log
[ $rc -ne 0 ] && exit $rc
python - <<'PY'
import sys; sys.path.insert(0, '.')
import bench, librectify_amd as L
ctx = L.Context(0)
print(bench.ransac_rates(ctx))
PY
bash tools/single_frame_trace.sh gpurun_out/sft_ransac 1 2>&1 | grep "filter_lines\|pencil\|ransac\|peel\|result_gather"
