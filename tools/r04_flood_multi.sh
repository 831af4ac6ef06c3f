#!/bin/bash
# multi-source re-walks: parity first, then what they buy (single frames, rounds, batch rate)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_source or flood or storage_tier or natural or full_size or carries or outgrows" > gpurun_out/r04_multi_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r04_multi_tests.log
[ $rc -ne 0 ] && exit $rc
for m in 0 1; do
  LIBRECTIFY_FLOOD_MULTI=$m timeout -k 10 300 python tools/run4k_seeds.py > gpurun_out/r04_multi_run4k_$m.txt 2>&1 || exit 1
  echo "== multi $m"; cat gpurun_out/r04_multi_run4k_$m.txt | tail -12
done
