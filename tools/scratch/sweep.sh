python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --batch1-iters 0 --pipeline 0 --streams 1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"stage_ms\"])"
python bench.py --steps 80 --warmup 8 --no-cpu-baseline --batch1-iters 0 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"])"
