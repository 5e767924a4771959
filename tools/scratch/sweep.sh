for r in 1 2; do for g in 0 1; do
echo -n "graphs=$g "
python bench.py --steps 80 --warmup 10 --no-cpu-baseline --batch1-iters 0 --graphs $g 2>/tmp/err | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' || tail -5 /tmp/err
done; done
