#!/bin/bash
# the whole GPU test suite, then the default bench line
mkdir -p gpurun_out
tag=${1:-x}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_tests_$tag.log 2>&1; rc=$?
tail -4 gpurun_out/r04_tests_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r04_bench_$tag.json 2> gpurun_out/r04_bench_$tag.err; echo "bench rc $?"
python3 - gpurun_out/r04_bench_$tag.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms/step", r["ms_per_step"], "h2d", r["h2d"]["GBps_per_rank"])
print("other", {k: v for k, v in r["other_rates_Mpix_per_s"].items() if k != "note"})
for k in ("single_frame", "flood", "natural_frame", "worst_case", "stage_ms_per_frame"):
    print(k, json.dumps(r.get(k))[:600])
print("roofline", r["roofline"]["frac"], r["roofline"]["kernel_ms"], "8k", r["roofline_8k"]["frac"])
print("ransac", {k: v["ms_per_solve"] for k, v in r["ransac_hypotheses_per_s"].items()})
print("cfg4", r["config4_batch1080"]["ms_per_pass"], "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["serial"]["value"])
PY
