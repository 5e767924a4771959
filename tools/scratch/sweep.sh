for c in 1 2; do for s in 2 3; do
echo -n "control=$c audio=$s "
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --batch1-iters 0 --control-streams $c --streams $s | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"])"
done; done
