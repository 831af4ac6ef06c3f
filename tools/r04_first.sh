#!/bin/bash
# round 4, first GPU call: the test suite, the single-rank RCCL rehearsal, the default bench line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04_tests_a.log 2>&1; echo "tests rc $?" 
tail -3 gpurun_out/r04_tests_a.log
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL timeout -k 10 600 python bench.py --gpus 1 --rehearse-collective --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/r04_rccl_world1.json 2> gpurun_out/r04_rccl_world1.err; echo "rccl rc $?"
tail -c 1500 gpurun_out/r04_rccl_world1.json
timeout -k 10 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err; echo "bench rc $?"
tail -c 600 gpurun_out/r04_bench_a.err
