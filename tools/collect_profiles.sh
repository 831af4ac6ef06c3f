#!/bin/bash
# On the GPU box: the measurements that go into profiles/ for a round.  usage: tools/collect_profiles.sh r05 [a|b]
# (two parts, each within one gpurun call: a = bench line, kernel statistics, roofline legs and HBM counters; b = flood rounds,
# single-frame timelines, the 8192^2 configuration, the slow-content traces)
R=${1:-r05}; PART=${2:-ab}
O=gpurun_out/profiles_$R
export TMPDIR=/tmp
mkdir -p $O
if [[ $PART == *a* ]]; then
python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/${R}_bench_under_rocprof.json 2> /dev/null
find $O/p1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${R}_kernel_stats.csv; rm -rf $O/p1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- python3 bench.py --roofline-only > $O/${R}_bench_roofline_leg.json 2> /dev/null
find $O/p2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${R}_kernel_stats_roofline_leg.csv; rm -rf $O/p2
W=8192 H=8192 NFRAMES=3 LAPS=4 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p3 -- python3 tools/bench_filter.py > $O/${R}_filter_8k.txt 2> /dev/null
find $O/p3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${R}_kernel_stats_filter_8k.csv; rm -rf $O/p3
bash tools/pmc_traffic.sh > $O/${R}_pmc_filter_traffic.txt 2>&1
bash tools/pmc_traffic.sh 8192 8192 3 > $O/${R}_pmc_filter_traffic_8k.txt 2>&1
rm -rf gpurun_out/pmc_* gpurun_out/cal_f gpurun_out/flt_*
fi
if [[ $PART == *b* ]]; then
bash tools/pmc_explore.sh > $O/${R}_pmc_flood_explore.txt 2>&1
rm -rf /tmp/pmcx_*
python3 tools/flood_debug.py bench > /dev/null 2> $O/${R}_flood_rounds_bench_frame.txt
python3 tools/run8k.py > $O/${R}_config5_8k.txt 2>&1
bash tools/single_frame_trace.sh $O/sft 1 2 3 4 > $O/${R}_single_frame_timeline.txt 2>&1; cp $O/sft/run.txt $O/${R}_single_frame.txt; cp $O/sft/kernel_stats.csv $O/${R}_kernel_stats_single_frames.csv; rm -rf $O/sft
python3 tools/single_call_sweep.py 8 > $O/${R}_single_call.txt 2>&1
python3 tools/bench_ransac.py > $O/${R}_ransac_cht_rates.txt 2>&1
for n in natural4k regions1080 radial1080 ramp4k diag40_1080; do bash tools/trace_frame.sh $n _$R > /dev/null 2>&1; (head -45 gpurun_out/trace_${n}_$R.txt; echo "..."; tail -4 gpurun_out/trace_${n}_$R.txt) > $O/${R}_trace_$n.txt; done
python3 tools/latency_fuzz.py > $O/${R}_latency_fuzz.txt 2>&1
python3 tools/run_mixed.py > $O/${R}_mixed_frames.txt 2>&1
fi
ls -la $O
