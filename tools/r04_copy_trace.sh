#!/bin/bash
# kernel + memory-copy trace of the batch call on host frames: when do a frame's transfer and its first kernel run?
kind=${1:-pinned}
export TMPDIR=/tmp
O=gpurun_out/ct_$kind
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/p -- python3 bench.py --steps 2 --warmup 1 --no-extra-legs --no-cpu-baseline --host-memory $kind > $O/run.json 2> $O/run.err
find $O/p -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace.csv
find $O/p -name "*memory_copy_trace.csv" | head -1 | xargs -I{} cp {} $O/memory_copy_trace.csv
rm -rf $O/p
python3 - $O <<'PY'
import csv, sys, collections
d = sys.argv[1]
K = list(csv.DictReader(open(d + "/kernel_trace.csv")))
C = list(csv.DictReader(open(d + "/memory_copy_trace.csv")))
print("copy columns:", list(C[0].keys()))
big = [c for c in C if int(c.get("Bytes", c.get("Size", 0)) or 0) >= 1 << 20] if ("Bytes" in C[0] or "Size" in C[0]) else C
print("copies:", len(C), "of >= 1 MB:", len(big))
big.sort(key=lambda c: int(c["Start_Timestamp"]))
t0 = int(big[len(big) // 3]["Start_Timestamp"])  # (skip the warm-up step)
sel = [c for c in big if int(c["Start_Timestamp"]) >= t0]
dur = [(int(c["End_Timestamp"]) - int(c["Start_Timestamp"])) / 1e3 for c in sel]
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
busy = sum(dur)
byt = sum(int(c.get("Bytes", c.get("Size", 0))) for c in sel)
print("large copies after warm-up: %d, mean duration %.1f us, link busy %.1f %% of the span, %.1f GB/s while busy, %.1f GB/s over the span"
      % (len(sel), sum(dur) / len(dur), 100 * busy / span, byt / busy / 1e3, byt / span / 1e3))
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(sel, sel[1:])]
gaps.sort()
print("gaps between consecutive large copies (us): median %.1f, p90 %.1f, max %.1f" % (gaps[len(gaps) // 2], gaps[int(len(gaps) * .9)], gaps[-1]))
# kernels: busy time per stream/queue
fil = [k for k in K if "filter_lanes" in k["Kernel_Name"] and int(k["Start_Timestamp"]) >= t0]
print("filter launches after warm-up:", len(fil))
PY
