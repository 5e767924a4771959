# Where HIP's hardware queues land on the command processor's pipes: the single-GPU pipeline timed with its streams first used in
# different orders (d = a dummy normal-priority stream, h = a dummy high-priority one)
export TMPDIR=/tmp
rm -rf gpurun_out/qo; mkdir -p gpurun_out/qo
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps 150 --warmup 10"
i=0
for o in ${ORDERS}; do
  i=$((i+1)); n=$(printf "%02d" $i)
  NWS_BENCH_QUEUE_ORDER=$o timeout 120 python bench.py $Q > gpurun_out/qo/${n}_${o//,/-}.json 2> gpurun_out/qo/$n.err
done
python - <<'PY'
import json, glob, os
for p in sorted(glob.glob("gpurun_out/qo/*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(f"{os.path.basename(p):50s} {d['ms_per_step']:.4f}")
    except Exception as e:
        print(p, "failed", e)
PY
