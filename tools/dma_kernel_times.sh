#!/bin/bash
# Which kernels of the batch get slower beside host-to-device DMA, and does the shader clock move?  (a) kernel trace of the
# headline batch with page-locked host frames and with resident frames: mean duration per kernel, lanes concurrent;
# (b) the same two under --pmc GRBM_GUI_ACTIVE (dispatches serialised by the profiler): duration and cycles per kernel,
# cycles / ns = the clock the kernel ran at.  usage: tools/dma_kernel_times.sh
export TMPDIR=/tmp
for mem in pinned device; do
  rm -rf /tmp/dkt_$mem /tmp/dkp_$mem
  timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dkt_$mem -- python3 bench.py --no-extra-legs --no-cpu-baseline --steps 3 --warmup 1 --host-memory $mem > /tmp/dkt_$mem.log 2>&1
  grep -o '"value": [0-9.]*' /tmp/dkt_$mem.log | head -1 | sed "s/^/$mem traced /"
  timeout -k 5 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/dkp_$mem -- python3 bench.py --no-extra-legs --no-cpu-baseline --steps 2 --warmup 1 --host-memory $mem > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,collections,re
def short(k):
    k=re.sub(r'\(anonymous namespace\)::','',k); return re.sub(r'<.*','',k.split('(')[0]).split('::')[-1]
tr={}
for mem in ("pinned","device"):
    f=glob.glob("/tmp/dkt_%s/*/*kernel_trace.csv" % mem); d=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])): d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    tr[mem]=d
print("(a) lanes concurrent: mean us per launch, page-locked host frames | resident frames | ratio   (launches, share of the sum)")
tot={m:sum(sum(v) for v in tr[m].values()) for m in tr}
for k in sorted(tr["pinned"], key=lambda k:-sum(tr["pinned"][k]))[:16]:
    a=tr["pinned"][k]; b=tr["device"].get(k,[0])
    ma=sum(a)/len(a)/1e3; mb=sum(b)/max(1,len(b))/1e3
    print("  %-30s %8.1f %8.1f  x%.3f   (%d, %.1f %%)" % (k,ma,mb,ma/max(mb,1e-9),len(a),100.0*sum(a)/tot["pinned"]))
print("  sum of kernel time per step-set: pinned %.1f ms, device %.1f ms, x%.3f" % (tot["pinned"]/1e6, tot["device"]/1e6, tot["pinned"]/tot["device"]))
print("(b) dispatches serialised (--pmc GRBM_GUI_ACTIVE): mean us | cycles/ns, pinned then device")
pm={}
for mem in ("pinned","device"):
    f=glob.glob("/tmp/dkp_%s/*/*counter_collection.csv" % mem); d=collections.defaultdict(lambda:[0,0.0,0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"]!="GRBM_GUI_ACTIVE": continue
        e=d[short(r["Kernel_Name"])]; e[0]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); e[1]+=float(r["Counter_Value"]); e[2]+=1
    pm[mem]=d
for k in sorted(pm["pinned"], key=lambda k:-pm["pinned"][k][0])[:10]:
    a=pm["pinned"][k]; b=pm["device"].get(k,[1,0,1])
    print("  %-30s %8.1f us %6.3f GHz | %8.1f us %6.3f GHz" % (k,a[0]/a[2]/1e3,a[1]/max(1,a[0]),b[0]/b[2]/1e3,b[1]/max(1,b[0])))
PY
rm -rf /tmp/dkt_* /tmp/dkp_*
