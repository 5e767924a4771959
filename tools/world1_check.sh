# Same-box A/B of the N > 1 issue pattern at world size 1: the single-GPU line against the forced-distributed
# lines (RCCL and copy-engine exchange, with and without sub-batches), host enqueue time per step beside each.
# Everything lands under gpurun_out/w1/.
export TMPDIR=/tmp
mkdir -p gpurun_out/w1
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps ${K:-200}"
run() {   # name, env..., -- args
    name=$1; shift
    envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout 150 python bench.py $Q "$@" > gpurun_out/w1/$name.json 2> gpurun_out/w1/$name.err
}
run single X=1 --
run rccl NWS_BENCH_FORCE_DIST=1 -- --gather rccl
run copy NWS_BENCH_FORCE_DIST=1 -- --gather copy
run rccl4 NWS_BENCH_FORCE_DIST=1 -- --gather rccl --gather-chunks 4
run copy4 NWS_BENCH_FORCE_DIST=1 -- --gather copy --gather-chunks 4
run queued NWS_BENCH_FORCE_DIST=1 NWS_BENCH_DIAG=queued -- --gather rccl
run single_b X=1 --
for extra in "$@"; do eval "$extra"; done
python - <<'PY'
import json, glob, os
base = None
for p in sorted(glob.glob("gpurun_out/w1/*.json"), key=os.path.getmtime):
    n = os.path.basename(p)[:-5]
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "failed", e, open(p[:-5] + ".err").read()[-600:]); continue
    if base is None:
        base = d["ms_per_step"]
    ex = d.get("exchange") or {}
    f = lambda v: "-" if v is None else f"{v:.4f}"
    print(f"{n:12s} ms/step {d['ms_per_step']:.4f}  x{d['ms_per_step']/base:.3f} of the first line  host_issue {d.get('host_issue_ms_per_step')}  "
          f"world1_overhead {f(ex.get('world1_overhead'))}  single_pattern {f(ex.get('single_gpu_pattern_ms'))}  compute_only {f(ex.get('compute_only_ms'))}  "
          f"gather {f(ex.get('gather_ms'))}  overlap_eff {f(ex.get('overlap_efficiency'))}  "
          f"selfcheck {(d.get('pipeline_selfcheck') or {}).get('mismatching_all_ranks', (d.get('pipeline_selfcheck') or {}).get('mismatching'))}")
PY
