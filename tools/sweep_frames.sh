for f in 64 128 256; do
echo "== frames $f"
GPU_MAX_HW_QUEUES=8 python3 bench.py --steps 6 --warmup 2 --frames $f --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
