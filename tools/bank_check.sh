# Exact sin-MLP shapers after the two-pass bank layout: the GPU tests that run them, then the exact bench line at the new occupancy
# (4 workgroups per CU) and with 35 KB of unused LDS per workgroup (= the two workgroups per CU the kernel had before).
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/bk
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "e2e or exact or bank or newt_and_fastnewt or full_size or batch64" > gpurun_out/bk/pytest_exact.txt 2>&1; tail -3 gpurun_out/bk/pytest_exact.txt
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --exact --steps 50"
timeout 120 python bench.py $Q > gpurun_out/bk/exact_new_1.json 2>/dev/null
NWS_EXCITER_BANK_LDS_PAD=35840 timeout 120 python bench.py $Q > gpurun_out/bk/exact_pad_1.json 2>/dev/null
timeout 120 python bench.py $Q > gpurun_out/bk/exact_new_2.json 2>/dev/null
NWS_EXCITER_BANK_LDS_PAD=35840 timeout 120 python bench.py $Q > gpurun_out/bk/exact_pad_2.json 2>/dev/null
NWS_EXCITER_BANK_LDS_PAD=12288 timeout 120 python bench.py $Q > gpurun_out/bk/exact_pad3wg.json 2>/dev/null
python - <<'PY'
import json
for n in ("exact_new_1", "exact_pad_1", "exact_new_2", "exact_pad_2", "exact_pad3wg"):
    try:
        d = json.loads(open(f"gpurun_out/bk/{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d.get("roofline", {}).get("kernel_ms"), d.get("stage_ms"))
    except Exception as e:
        print(n, "failed", e)
PY
