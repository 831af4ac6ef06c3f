#!/bin/bash
# on the GPU box: rebuild the library with variants of the filter kernel and time each (tools/bench_filter.py)
# usage: tools/try_filter_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...
mkdir -p gpurun_out/filter_variants
for v in "$@"; do
  LR_FILTER_FLAGS="-fno-honor-nans -mno-amdgpu-ieee $v" python3 -m librectify_amd.build --force > /dev/null 2>&1
  echo "== $v"; python3 tools/bench_filter.py 2>/dev/null; W=8192 H=8192 NFRAMES=3 python3 tools/bench_filter.py 2>/dev/null
done
python3 -m librectify_amd.build --force > /dev/null 2>&1
