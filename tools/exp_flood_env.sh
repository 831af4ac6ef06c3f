#!/bin/bash
# on the GPU box: throughput of bench.py's timed region under environment settings of the library's knobs.
# usage: [MEM=pinned|pageable] tools/exp_flood_env.sh "VAR=val ..." "VAR=val ..." ...   (X=1 for the defaults)
for e in "$@"; do
  echo "== $e"; env $e GPU_MAX_HW_QUEUES=8 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-legs --host-memory ${MEM:-pinned} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_frame'])"
done
