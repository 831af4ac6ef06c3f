#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/sft_regions; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 tools/run_regions1080.py > $O/run.txt 2>/dev/null
find $O/p -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace.csv; rm -rf $O/p
python3 - "$O/kernel_trace.csv" > $O.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(s):
    s = s.replace('(anonymous namespace)::', '')
    s = re.sub(r'<.*', '', s.split('(')[0])
    return s.split('::')[-1][-34:]
last = max(i for i, r in enumerate(rows) if 'filter_lanes' in r['Kernel_Name'])
t0 = int(rows[last]['Start_Timestamp']); prev_end = t0
for r in rows[last:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {nm(r['Kernel_Name'])}  grid {r.get('Grid_Size_X', '?')}")
    prev_end = e
PY
