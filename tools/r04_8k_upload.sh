#!/bin/bash
# BASELINE config 5 (one 8192x8192 frame) from a pageable and from a page-locked buffer, 8 and 16 staging threads;
# then the single-frame and batch keys without a NUMA binding of the process (the library binds its own helpers)
python tools/upload_8k.py 8 2>&1 | grep num_threads
python tools/upload_8k.py 16 2>&1 | grep num_threads
LIBRECTIFY_STAGING_BIND=0 python tools/upload_8k.py 16 2>&1 | grep num_threads | sed 's/^/(helpers unbound) /'
for b in 1 0; do
LIBRECTIFY_STAGING_BIND=$b python bench.py --steps 6 --warmup 2 --no-extra-legs --no-cpu-baseline --numa none 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('process unbound, helpers bound=$b: value %8.1f Mpix/s  h2d %s' % (r['value'], (r.get('h2d') or {}).get('GBps_per_rank')))"
done
