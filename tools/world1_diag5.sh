export TMPDIR=/tmp
rm -rf gpurun_out/w1h; mkdir -p gpurun_out/w1h
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps 200"
i=0
for o in x,a0,a1,c0,c1 a0,a1,x,c0,d,d,c1 a0,a1,c0,c1,d,d,x a0,a1,c0,x,d,d,d,c1 x,a0,a1,c0,c1; do
 for k in blit3 rccl copy; do
  i=$((i+1)); n=$(printf "%02d" $i)
  d=""; g=$k; [ $k = blit3 ] && { d=blit3; g=rccl; }
  NWS_BENCH_QUEUE_ORDER=$o NWS_BENCH_DIAG=$d NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --gather $g > gpurun_out/w1h/${n}_${k}_${o//,/-}.json 2> gpurun_out/w1h/$n.err
 done
done
python - <<'PY'
import json, glob, os
for p in sorted(glob.glob("gpurun_out/w1h/*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        ex = d.get("exchange") or {}
        print(f"{os.path.basename(p):44s} {d['ms_per_step']:.4f} w1 {ex.get('world1_overhead'):.4f} plain {ex.get('single_gpu_pattern_ms'):.4f} compute_only {ex.get('compute_only_ms'):.4f} gather {ex.get('gather_ms'):.4f}")
    except Exception as e:
        print(p, "failed", e)
PY
