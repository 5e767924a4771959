for r in 1 2; do for q in 4 8; do
echo -n "hwq=$q "
GPU_MAX_HW_QUEUES=$q python bench.py --steps 80 --warmup 8 --no-cpu-baseline --batch1-iters 0 2>/tmp/err | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("plain", d["ms_per_step"])' || tail -5 /tmp/err
done; done
