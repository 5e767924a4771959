# End-of-round check of the final tree: runtime-size path A/B of the tile rule, the whole GPU suite, the bench lines and the
# one-stream kernel statistics.  Steps ordered by importance; everything lands under gpurun_out/fc2/.
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/fc2
NWS_G_TILE_LDS=81920 timeout 100 python tools/generic_profile.py 64 500 1 500 > gpurun_out/fc2/generic_old_rule.txt 2>&1
timeout 100 python tools/generic_profile.py 64 500 1 500 > gpurun_out/fc2/generic_new_rule.txt 2>&1
grep -h "runtime-size" gpurun_out/fc2/generic_old_rule.txt gpurun_out/fc2/generic_new_rule.txt
timeout 420 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/fc2/pytest_gpu.txt 2>&1; tail -2 gpurun_out/fc2/pytest_gpu.txt
timeout 200 python bench.py > gpurun_out/fc2/bench_default.json 2> gpurun_out/fc2/bench_default.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/fc2/bench_driver_k20.json 2>/dev/null
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0"
timeout 100 python bench.py $Q --inputs realistic > gpurun_out/fc2/bench_realistic_inputs.json 2>/dev/null
timeout 100 python bench.py $Q --exciter-opts 8 > gpurun_out/fc2/bench_hybrid_w_optin.json 2>/dev/null
timeout 100 python bench.py $Q --exact --steps 50 > gpurun_out/fc2/bench_exact_shapers.json 2>/dev/null
timeout 400 bash tools/collect_profiles.sh r04 > gpurun_out/fc2/collect.log 2>&1
python - <<'PY'
import json
for n in ("bench_default", "bench_driver_k20", "bench_realistic_inputs", "bench_hybrid_w_optin", "bench_exact_shapers"):
    try:
        d = json.loads(open(f"gpurun_out/fc2/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d.get("stage_ms"))
    except Exception as e:
        print(n, "failed", e)
PY
head -6 gpurun_out/prof_r04/rocprofv3_summary.txt | cut -c1-200
