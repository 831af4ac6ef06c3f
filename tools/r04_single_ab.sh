#!/bin/bash
# whole single calls (wall clock through the reference's entry): logs x just-in-time rounds
for cfg in "0 0" "1 0" "1 1" "0 1" "1 1" "1 0"; do
  set -- $cfg
  echo "== LOGS=$1 JIT=$2"
  LIBRECTIFY_FLOOD_LOGS=$1 LIBRECTIFY_FLOOD_JIT=$2 timeout -k 10 120 python tools/single_call_sweep.py 8 2>&1 | tail -2
done
