export TMPDIR=/tmp
rm -rf gpurun_out/w1f; mkdir -p gpurun_out/w1f
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps 200"
D() { n=$1; d=$2; shift 2; NWS_BENCH_DIAG=$d NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q "$@" > gpurun_out/w1f/$n.json 2> gpurun_out/w1f/$n.err; }
D 01_rccl wprof,plainfirst --gather rccl
D 02_rccl_noexch wprof,noexch,plainfirst --gather rccl
D 03_rccl_blit3 wprof,blit3,plainfirst --gather rccl
D 04_copy wprof,plainfirst --gather copy
D 05_rccl_b wprof,plainfirst --gather rccl
D 06_rccl_blit3_b wprof,blit3,plainfirst --gather rccl
python - <<'PY'
import json, glob, os
for p in sorted(glob.glob("gpurun_out/w1f/*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        ex = d.get("exchange") or {}
        print(f"{os.path.basename(p):24s} {d['ms_per_step']:.4f} w1 {ex.get('world1_overhead'):.4f} plain {ex.get('single_gpu_pattern_ms'):.4f} plain_first {d.get('plain_first_ms'):.4f} compute_only {ex.get('compute_only_ms'):.4f} gather {ex.get('gather_ms'):.4f} host {d.get('host_issue_ms_per_step')} worker {d.get('exchange_worker_us')}")
    except Exception as e:
        print(p, "failed", e, open(p[:-5] + ".err").read()[-400:])
PY
