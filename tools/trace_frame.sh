#!/bin/bash
# Kernel timeline of the LAST call of tools/run_frame.py <name> under rocprofv3 --kernel-trace, and the time per kernel name.
# usage: tools/trace_frame.sh <name> [tag]   -> gpurun_out/trace_<name><tag>.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
N=$1; O=gpurun_out/trace_$N$2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 tools/run_frame.py $N 3 > $O/run.txt 2>/dev/null
find $O/p -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace.csv; rm -rf $O/p
python3 - "$O/kernel_trace.csv" > $O.txt <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(s):
    s = s.replace('(anonymous namespace)::', '')
    s = re.sub(r'<.*', '', s.split('(')[0])
    return s.split('::')[-1][-34:]
last = max(i for i, r in enumerate(rows) if 'filter_lanes' in r['Kernel_Name'])
t0 = int(rows[last]['Start_Timestamp']); prev_end = t0
tot = collections.OrderedDict()
lines = []
for r in rows[last:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    k = nm(r['Kernel_Name'])
    lines.append(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {k}  grid {r.get('Grid_Size_X', '?')}")
    a = tot.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    prev_end = max(prev_end, e)
print(f"last call: {len(rows) - last} launches, {(prev_end - t0) / 1e3:.1f} us from the filter's start to the last kernel's end")
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {us:10.1f} us  {n:5d} x  {k}")
print()
print("\n".join(lines[:400]))
PY
cat $O/run.txt >> $O.txt
rm -rf $O
