export TMPDIR=/tmp
rm -rf gpurun_out/w1g; mkdir -p gpurun_out/w1g
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps 200"
i=0
for o in "" x,a0,a1,c0,c1 a0,x,a1,c0,c1 a0,a1,x,c0,c1 a0,a1,c0,c1,x a0,a1,c0,c1,d,x a0,a1,c0,c1,d,d,x a0,a1,c0,c1,d,d,d,x a0,c0,a1,c1,x d,d,d,a0,a1,c0,c1,x; do
  i=$((i+1)); n=$(printf "%02d" $i)
  NWS_BENCH_QUEUE_ORDER=$o NWS_BENCH_DIAG=blit3 NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --gather rccl > gpurun_out/w1g/${n}_blit3_${o//,/-}.json 2> gpurun_out/w1g/$n.err
done
python - <<'PY'
import json, glob, os
for p in sorted(glob.glob("gpurun_out/w1g/*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        ex = d.get("exchange") or {}
        print(f"{os.path.basename(p):44s} {d['ms_per_step']:.4f} w1 {ex.get('world1_overhead'):.4f} plain {ex.get('single_gpu_pattern_ms'):.4f} compute_only {ex.get('compute_only_ms'):.4f} gather {ex.get('gather_ms'):.4f}")
    except Exception as e:
        print(p, "failed", e)
PY
