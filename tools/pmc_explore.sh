#!/bin/bash
# usage: tools/pmc_explore.sh   (run on the GPU box via gpurun) -- instruction mix of the flood exploration kernel,
# per dispatch (the first seven rounds of the last of the three frames of tools/run4k.py)
export TMPDIR=/tmp
cat <<'HDR'
# tools/pmc_explore.sh on an MI355X box (rocprofv3 --pmc, two passes), tools/run4k.py: one 3840x2160 synthetic frame,
# single-frame call.  Columns = consecutive flood_explore_kernel dispatches of the last frame: the first is an empty
# launch past the end of the previous frame's flood (rounds are enqueued blindly), then rounds 1..6 (40 611 seeds).
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count in units of four cycles.
HDR
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcx_$i -- python3 tools/run4k.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmcx_$i/*/*counter_collection.csv")[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "flood_explore_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    per = len(v) // 3  # three frames in tools/run4k.py: the last frame's rounds (the ones past the end of the flood are empty launches)
    print("%-22s" % k, " ".join("%13.0f" % x for x in v[-per:][:7]))
PY
done
