#!/bin/bash
# the label tests under the fall-back configurations: blind rounds, no logs, no giants' rule
mkdir -p gpurun_out
for v in "LIBRECTIFY_FLOOD_JIT=0" "LIBRECTIFY_FLOOD_LOGS=0" "LIBRECTIFY_FLOOD_GIANTS=0" "LIBRECTIFY_FLOOD_DENSE_DIV=0 LIBRECTIFY_FLOOD_JIT_LANES=0"; do
  echo "== $v"
  env $v timeout -k 10 900 python -m pytest tests -m gpu -q -k "not rewalks_from_the_logs and not giant_walks and not outgrows_its_grid" 2>&1 | tail -2
done
