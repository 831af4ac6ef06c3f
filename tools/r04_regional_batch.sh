#!/bin/bash
# the early hand-over to the second tier (frames with many long walks) in the lanes of a batch: resident frames
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs --host-memory device 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f Mpix/s' % d['value'])
"
}
for rep in 1 2; do
echo "== as it is (hand-over at 32 tiles once 16 walks outgrew the first tier)"; run
echo "== never early"; LIBRECTIFY_FLOOD_T1_REGIONAL_MIN=1000000 run
echo "== early at 64 tiles"; LIBRECTIFY_FLOOD_T1_REGIONAL=64 run
echo "== early at 96 tiles"; LIBRECTIFY_FLOOD_T1_REGIONAL=96 run
done
