#!/bin/bash
# HBM traffic of the filter kernel per launch, with FETCH_SIZE calibrated on a known read of the same
# access shape (4 B/lane).  Run on the GPU box; prints a small report (copy it to profiles/).
# usage: tools/pmc_traffic.sh [width height frames]   (default 3840 2160 10; 8192 8192 3 for the survey's primary size)
PW=${1:-3840}; PH=${2:-2160}; PN=${3:-10}
export TMPDIR=/tmp
rd() { python3 - "$1" "$2" "$3" <<'PY'
import csv,glob,sys
d,kern,ctr=sys.argv[1:4]
f=glob.glob(d+"/*/*counter_collection.csv")[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"]==ctr]
print(sum(v)/len(v))
PY
}
test -x tools/ubench/read_calib || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/read_calib tools/ubench/read_calib.hip
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/cal_f -- tools/ubench/read_calib > /dev/null 2>&1
CAL=$(rd gpurun_out/cal_f read4 FETCH_SIZE)
rm -rf gpurun_out/flt_f gpurun_out/flt_w
W=$PW H=$PH LAPS=2 NFRAMES=$PN rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/flt_f -- python3 tools/bench_filter.py > /dev/null 2>&1
W=$PW H=$PH LAPS=2 NFRAMES=$PN rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/flt_w -- python3 tools/bench_filter.py > /dev/null 2>&1
F=$(rd gpurun_out/flt_f filter_ FETCH_SIZE); W=$(rd gpurun_out/flt_w filter_ WRITE_SIZE)
python3 - <<PY
cal=$CAL; f=$F; w=$W; pw=$PW; ph=$PH
true_kb = 2*1024*1024  # 2 GiB read by read4, in KiB
corr = true_kb / cal
print("calibration: read4 FETCH_SIZE = %.0f KiB for %.0f KiB read -> correction x%.3f" % (cal, true_kb, corr))
print("filter kernel per launch (%dx%d): FETCH_SIZE %.0f KiB (corrected %.1f MB), WRITE_SIZE %.0f KiB (%.1f MB)" % (pw, ph, f, f*corr*1024/1e6, w, w*1024/1e6))
print("traffic_bytes_per_launch %.0f" % ((f*corr + w)*1024))
print("algorithmic_bytes_per_launch %.0f (18 B/px)" % (18*pw*ph))
PY
