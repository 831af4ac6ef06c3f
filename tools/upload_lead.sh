for m in pinned pageable; do
LIBRECTIFY_LANE_DEBUG=1 GPU_MAX_HW_QUEUES=8 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --host-memory $m > gpurun_out/r02u/lane_$m.json 2> gpurun_out/r02u/lane_$m.txt
python3 - gpurun_out/r02u/lane_$m.txt gpurun_out/r02u/lane_$m.json $m <<'PY'
import re,sys,json
lead=[float(x) for x in re.findall(r"upload done (-?[\d.]+) ms", open(sys.argv[1]).read())]
lead=lead[-64:]
print(sys.argv[3], "value", json.load(open(sys.argv[2]))["value"], "frames that waited for their upload (lead < 0.2 ms):", sum(1 for x in lead if x < 0.2), "of", len(lead), " median lead %.2f" % sorted(lead)[len(lead)//2])
PY
done
