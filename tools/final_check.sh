# One short GPU call at the end of a round: the GPU suite on the tree as it is, a same-box A/B of the XCD-aware launch orders
# (NWS_EXCITER_XCD / NWS_MLP_XCD = 0 restore the old ones), the driver's bench command with counters, kernel stats.
# Steps are ordered by importance; each writes under gpurun_out/fc/ as it finishes.
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/fc
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0"
timeout 420 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/fc/pytest_gpu.txt 2>&1; tail -3 gpurun_out/fc/pytest_gpu.txt
NWS_EXCITER_XCD=0 NWS_MLP_XCD=0 timeout 120 python bench.py $Q > gpurun_out/fc/ab_old_1.json 2>/dev/null
timeout 120 python bench.py $Q > gpurun_out/fc/ab_new_1.json 2>/dev/null
NWS_EXCITER_XCD=0 NWS_MLP_XCD=0 timeout 120 python bench.py $Q > gpurun_out/fc/ab_old_2.json 2>/dev/null
timeout 120 python bench.py $Q > gpurun_out/fc/ab_new_2.json 2>/dev/null
python - <<'PY'
import json
for n in ("ab_old_1", "ab_new_1", "ab_old_2", "ab_new_2"):
    try:
        d = json.loads(open(f"gpurun_out/fc/{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d.get("roofline", {}).get("kernel_ms"), {k: v.get("ms") for k, v in d.get("roofline_all", {}).items()} if isinstance(d.get("roofline_all"), dict) else "")
    except Exception as e:
        print(n, "failed", e)
PY
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/fc/bench_driver_k20.json 2> gpurun_out/fc/bench_driver_k20.err; tail -c 600 gpurun_out/fc/bench_driver_k20.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fc/trace_1stream -- python bench.py $Q --warmup 5 --steps 30 --pipeline 0 --streams 1 > gpurun_out/fc/trace_1stream.log 2>&1
find gpurun_out/fc/trace_1stream -name "*kernel_stats.csv" -exec cp {} gpurun_out/fc/rocprofv3_kernel_stats_1stream.csv \;
find gpurun_out/fc -name "*.db" -delete; find gpurun_out/fc -name "*kernel_trace.csv" -delete
head -12 gpurun_out/fc/rocprofv3_kernel_stats_1stream.csv | cut -c1-160
