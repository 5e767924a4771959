# fir_noise_mfma_kernel after the rotated copies: parity, one-stream time, LDS bank conflicts (PMC pass of its own)
export TMPDIR=/tmp
mkdir -p gpurun_out/fir
[ -n "$SKIP_TESTS" ] || { timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "noise or e2e or streamed or stream" > gpurun_out/fir/pytest.txt 2>&1; tail -1 gpurun_out/fir/pytest.txt; }
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0"
timeout 120 python bench.py $Q --steps 200 > gpurun_out/fir/pipe.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/fir/pipe.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], d['stage_ms'])"
cd /tmp
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/fir/pmc -- python $GRAFT_REPO_ROOT/bench.py $Q --steps 6 --warmup 2 --pipeline 0 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/fir/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob("gpurun_out/fir/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        import re
        k = re.sub(r"[<(].*", "", re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if v.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and ("fir_noise" in k or "exciter" in k or "frame_mlps" in k or "col125" in k):
        print(f"{k:60s} bank_conflict/idx_active {v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.3f}")
PY
find gpurun_out/fir/pmc -name "*.csv" -size +4M -delete
