#!/bin/bash
# the batch (bench.py's headline leg only) with the logs off and on at several thresholds
mkdir -p gpurun_out
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s, %.2f ms per step' % (d['value'], d['ms_per_step']))
"
}
echo "== logs off"; LIBRECTIFY_FLOOD_LOGS=0 run
for cfg in "$@"; do
  set -- $cfg
  echo "== logs on, LOG_MIN=$1 LOG_WALK=$2"
  LIBRECTIFY_FLOOD_LOGS=1 LIBRECTIFY_FLOOD_LOG_MIN=$1 LIBRECTIFY_FLOOD_LOG_WALK=$2 run
done
echo "== logs off"; LIBRECTIFY_FLOOD_LOGS=0 run
