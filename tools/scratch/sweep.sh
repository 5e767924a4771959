for c in 1 2; do
echo -n "per-utterance control=$c "
python bench.py --steps 80 --warmup 8 --no-cpu-baseline --batch1-iters 0 --control-streams $c --gru per-utterance | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["gru_ms_in_timed_region"])'
done
