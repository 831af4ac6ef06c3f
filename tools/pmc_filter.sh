#!/bin/bash
# usage: tools/pmc_filter.sh <kernel-name-substring>   (run on the GPU box via gpurun)
export TMPDIR=/tmp
K=${1:-filter}
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  LAPS=2 NFRAMES=10 timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$i -- python3 tools/bench_filter.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$i/*/*counter_collection.csv")[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "$K" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("%-24s %16.1f  (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
done
