run() { for i in 1 2 3 4 5 6 7 8; do python -m pytest tests -m gpu -q -p no:cacheprovider -k "pipeline" 2>&1 | grep -E "passed|failed" | cut -c1-20; done | sort | uniq -c; }
echo "HPB2 no scratch:"; run
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --batch1-iters 0 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"], d[\"stage_ms\"])"
