export TMPDIR=/tmp
rm -rf gpurun_out/w1e; mkdir -p gpurun_out/w1e
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps 200"
r() { n=$1; shift; env "$@" > /dev/null 2>&1; }
timeout 120 python bench.py $Q > gpurun_out/w1e/00_single.json 2>/dev/null
timeout 120 python bench.py $Q --control-streams 1 > gpurun_out/w1e/01_single_cs1.json 2>/dev/null
NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --gather rccl --control-streams 1 > gpurun_out/w1e/02_rccl_cs1.json 2>/dev/null
NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --gather copy --control-streams 1 > gpurun_out/w1e/03_copy_cs1.json 2>/dev/null
NWS_BENCH_DIAG=lag NWS_BENCH_LAG=3 NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --gather rccl > gpurun_out/w1e/04_rccl_lag3.json 2>/dev/null
NWS_BENCH_DIAG=lag NWS_BENCH_LAG=4 NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --gather rccl --depth 6 > gpurun_out/w1e/05_rccl_lag4_d6.json 2>/dev/null
NWS_BENCH_DIAG=lag NWS_BENCH_LAG=6 NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --gather rccl --depth 8 > gpurun_out/w1e/06_rccl_lag6_d8.json 2>/dev/null
NWS_BENCH_DIAG=lag,fake NWS_BENCH_LAG=4 NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --gather rccl --depth 6 > gpurun_out/w1e/07_fake_lag4_d6.json 2>/dev/null
timeout 120 python bench.py $Q --depth 6 > gpurun_out/w1e/08_single_d6.json 2>/dev/null
NWS_BENCH_DIAG=lag NWS_BENCH_LAG=4 NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --gather copy --depth 6 > gpurun_out/w1e/09_copy_lag4_d6.json 2>/dev/null
python - <<'PY'
import json, glob, os
base = None
for p in sorted(glob.glob("gpurun_out/w1e/*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        base = base or d["ms_per_step"]
        ex = d.get("exchange") or {}
        print(f"{os.path.basename(p):36s} {d['ms_per_step']:.4f} x{d['ms_per_step']/base:.3f} host {d.get('host_issue_ms_per_step')} compute_only {ex.get('compute_only_ms')} {(d.get('pipeline_selfcheck') or {}).get('mismatching_all_ranks')}")
    except Exception as e:
        print(p, "failed", e)
PY
