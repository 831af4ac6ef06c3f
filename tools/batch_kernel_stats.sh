#!/bin/bash
# On the GPU box: per-kernel averages inside the six-lane batch (bench.py under rocprofv3 --kernel-trace --stats), the
# flood's kernels first.  usage: tools/batch_kernel_stats.sh <outdir>
O=${1:-gpurun_out/bks}
export TMPDIR=/tmp
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench.json 2> /dev/null
find $O/p -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/p
python3 - $O <<'PY'
import csv, json, sys
O = sys.argv[1]
rows = list(csv.DictReader(open(O + '/kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]:
    n = r['Name'].split('(')[0].split('::')[-1]
    print("%-34s calls %6s avg %8.1f us  %5.1f %%" % (n[:34], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print("value under rocprof:", json.loads(open(O + '/bench.json').read().strip().splitlines()[-1])["value"])
PY
