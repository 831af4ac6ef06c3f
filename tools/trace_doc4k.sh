#!/bin/bash
# On the GPU box: kernel timeline of the natural 4K frame (tools/run_doc4k.py) under rocprofv3 --kernel-trace; prints the
# last frame's kernels.  usage: tools/trace_doc4k.sh <outdir>
O=${1:-gpurun_out/doc4k}
export TMPDIR=/tmp
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 tools/run_doc4k.py > $O/run.txt 2>/dev/null
find $O/p -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace.csv
rm -rf $O/p
python3 - "$O/kernel_trace.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(s):
    s = s.replace('(anonymous namespace)::', '')
    s = re.sub(r'<.*', '', s.split('(')[0])
    return s.split('::')[-1][-34:]
firsts = [i for i, r in enumerate(rows) if 'seed_count' in r['Kernel_Name']]
last = firsts[-1]
t0 = int(rows[last]['Start_Timestamp']); prev_end = t0
for r in rows[last:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {nm(r['Kernel_Name'])}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
    prev_end = e
PY
cat $O/run.txt
