#!/bin/bash
export LIBRECTIFY_FLOOD_HOLD=50 LIBRECTIFY_FLOOD_HOLD_START=1
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "== LOGS=$1 JIT=$2"
  LIBRECTIFY_FLOOD_LOGS=$1 LIBRECTIFY_FLOOD_JIT=$2 timeout -k 10 120 python tools/run4k_seeds.py 1 2>&1 | tail -2 | cut -c1-330
done
