#!/bin/bash
# Does host-to-device DMA beside the kernels lengthen their memory reads?  Average read latency at the L2's memory interface
# (TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum, L2 cycles) and DRAM-read credit stalls per kernel of the batch, with the
# frames coming from page-locked host memory (DMA running) and already resident (no DMA).  usage: tools/pmc_dma_latency.sh
export TMPDIR=/tmp
for mem in pinned device; do
  i=0
  for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); rm -rf /tmp/pmcd_$i
    timeout -k 5 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcd_$i -- python3 bench.py --no-extra-legs --no-cpu-baseline --steps 2 --warmup 1 --host-memory $mem > /dev/null 2>&1
    python3 - $mem $i <<'PY'
import csv,glob,collections,sys,re
mem,i=sys.argv[1],sys.argv[2]
f=glob.glob("/tmp/pmcd_%s/*/*counter_collection.csv" % i)
if not f: print(mem, "no counters collected"); sys.exit(0)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=re.sub(r'\(anonymous namespace\)::','',r["Kernel_Name"]); k=re.sub(r'<.*','',k.split('(')[0]).split('::')[-1]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k in ("flood_explore_kernel","flood_rewalk_kernel","filter_lanes_kernel","fit_kernel","component_scatter_kernel","flood_commit_pixels_kernel"):
    if k in acc:
        v=acc[k]; names=sorted(v)
        line="%-7s %-28s" % (mem,k) + "  ".join("%s %.4g" % (c, v[c]/max(1,n[(k,c)])) for c in names)
        if "TCC_EA0_RDREQ_LEVEL_sum" in v and v.get("TCC_EA0_RDREQ_sum",0)>0: line+="   -> mean read latency %.0f L2 cycles" % (v["TCC_EA0_RDREQ_LEVEL_sum"]/v["TCC_EA0_RDREQ_sum"])
        if "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" in v and v.get("TCC_EA0_RDREQ_DRAM_sum",0)>0: line+="   -> credit stalls per DRAM read %.3f" % (v["TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"]/v["TCC_EA0_RDREQ_DRAM_sum"])
        if "TCC_HIT_sum" in v: line+="   -> L2 hit rate %.4f" % (v["TCC_HIT_sum"]/max(1.0,v["TCC_HIT_sum"]+v["TCC_MISS_sum"]))
        print(line)
PY
  done
done
rm -rf /tmp/pmcd_*
