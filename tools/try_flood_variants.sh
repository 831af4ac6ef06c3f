#!/bin/bash
# on the GPU box: rebuild the library with -D variants (applied to every file) and time single frames + throughput
# usage: tools/try_flood_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...
for v in "$@"; do
  LR_EXTRA_FLAGS="$v" python3 -m librectify_amd.build --force > /dev/null 2>&1
  echo "== $v"; python3 tools/run4k_seeds.py 1 4 2>/dev/null
  GPU_MAX_HW_QUEUES=8 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --host-memory pinned 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('throughput (pinned host)', d['value'])"
done
python3 -m librectify_amd.build --force > /dev/null 2>&1
