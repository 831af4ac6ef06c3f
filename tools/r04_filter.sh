#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "filter or seeds_match or full_path or golden or strides or degenerate or random_small or doc_image or full_size or baseline" > gpurun_out/r04_filter_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r04_filter_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py --roofline-only 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('roofline', json.dumps({k: r['roofline'][k] for k in ('achieved','frac','kernel_ms','frac_moved_bytes')}))"
python tools/bench_filter.py 2>/dev/null | tail -3
