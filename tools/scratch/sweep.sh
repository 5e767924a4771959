for x in 0 1; do
if [ $x = 1 ]; then export NWS_X_SAMEFRAG=1; fi
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --batch1-iters 0 --pipeline 0 --streams 1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"stage_ms\"])"
done
