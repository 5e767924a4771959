# Bisecting the N > 1 issue pattern at world size 1 (NWS_BENCH_DIAG switches of bench.py), same box.
export TMPDIR=/tmp
rm -rf gpurun_out/w1d; mkdir -p gpurun_out/w1d
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps 200"
timeout 120 python bench.py $Q > gpurun_out/w1d/00_single.json 2>/dev/null
i=1
for d in ${DIAGS:-"" fake fake,hi fake,evaudio fake,recordonly fake,lag lag}; do
  [ "$d" = "-" ] && d=""
  n=$(printf "%02d" $i)_rccl_${d//,/_}
  NWS_BENCH_DIAG=$d NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --gather rccl > gpurun_out/w1d/$n.json 2> gpurun_out/w1d/$n.err
  i=$((i+1))
done
python - <<'PY'
import json, glob, os
base = None
for p in sorted(glob.glob("gpurun_out/w1d/*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        base = base or d["ms_per_step"]
        ex = d.get("exchange") or {}
        print(f"{os.path.basename(p):44s} {d['ms_per_step']:.4f} x{d['ms_per_step']/base:.3f} host {d.get('host_issue_ms_per_step')} compute_only {ex.get('compute_only_ms')} {d.get('pipeline_selfcheck')}")
    except Exception as e:
        print(p, "failed", e, open(p[:-5] + ".err").read()[-400:] if os.path.exists(p[:-5] + ".err") else "")
PY
