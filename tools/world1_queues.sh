# Which hardware queues the pipeline's streams land on with and without an RCCL communicator, and what more queues buy.
export TMPDIR=/tmp
mkdir -p gpurun_out/w1q
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0"
cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/w1q/trace_single -- python $R/bench.py $Q --steps 40 > $R/gpurun_out/w1q/trace_single.json 2> $R/gpurun_out/w1q/trace_single.err
NWS_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/w1q/trace_rccl -- python $R/bench.py $Q --steps 40 --gather rccl > $R/gpurun_out/w1q/trace_rccl.json 2> $R/gpurun_out/w1q/trace_rccl.err
cd $R
python tools/queue_map.py gpurun_out/w1q/trace_single > gpurun_out/w1q/queues_single.txt 2>&1
python tools/queue_map.py gpurun_out/w1q/trace_rccl > gpurun_out/w1q/queues_rccl.txt 2>&1
for q in 4 8 12 16 24; do
  GPU_MAX_HW_QUEUES=$q NWS_BENCH_FORCE_DIST=1 timeout 120 python bench.py $Q --steps 200 --gather rccl > gpurun_out/w1q/rccl_q$q.json 2>/dev/null
  GPU_MAX_HW_QUEUES=$q timeout 120 python bench.py $Q --steps 200 > gpurun_out/w1q/single_q$q.json 2>/dev/null
done
NWS_BENCH_INIT_PG_ONLY=1 timeout 120 python bench.py $Q --steps 200 > gpurun_out/w1q/single_pgonly.json 2>/dev/null
python - <<'PY'
import json, glob, os
for p in sorted(glob.glob("gpurun_out/w1q/*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        ex = d.get("exchange") or {}
        print(f"{os.path.basename(p):24s} {d['ms_per_step']:.4f} host {d.get('host_issue_ms_per_step')} compute_only {ex.get('compute_only_ms')}")
    except Exception as e:
        print(p, "failed", e)
PY
# keep the traces small
find gpurun_out/w1q -name "*.csv" -size +20M -delete
