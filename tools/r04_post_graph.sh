#!/bin/bash
# the stages behind the flood as one graph launch: parity first, then whole single calls and the batch, off / on
set -o pipefail
mkdir -p gpurun_out
LIBRECTIFY_POST_GRAPH=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_tests_graph.txt 2>&1; rc=$?
tail -3 gpurun_out/r04_tests_graph.txt
[ $rc -ne 0 ] && exit $rc
for g in 0 1 0 1; do
  echo "== LIBRECTIFY_POST_GRAPH=$g"
  LIBRECTIFY_POST_GRAPH=$g timeout -k 10 120 python tools/single_call_sweep.py 8 2>&1 | tail -2
  LIBRECTIFY_POST_GRAPH=$g timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   batch %.0f Mpix/s' % d['value'])
"
done
