# Range-proven lookups of the fused tail (NwsWeights.exciter_bound) against the clamped form (NWS_EXCITER_NO_RANGE=1): parity subset,
# then same-box A/B of the one-stream kernel time, the pipelined step and the realistic-input step
export TMPDIR=/tmp
mkdir -p gpurun_out/rg
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "range_proven" > gpurun_out/rg/pytest.txt 2>&1; tail -1 gpurun_out/rg/pytest.txt
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0"
for i in 1 2; do
  for nr in 1 ""; do
    NWS_EXCITER_NO_RANGE=$nr timeout 120 python bench.py $Q --steps 200 > gpurun_out/rg/pipe_$nr.json 2>/dev/null
    NWS_EXCITER_NO_RANGE=$nr timeout 120 python bench.py $Q --steps 200 --inputs realistic > gpurun_out/rg/real_$nr.json 2>/dev/null
    python - <<PY
import json
p=json.loads(open('gpurun_out/rg/pipe_$nr.json').read().strip().splitlines()[-1]); r=json.loads(open('gpurun_out/rg/real_$nr.json').read().strip().splitlines()[-1])
print('no_range', '$nr' or 0, 'one-stream exciter ms', p['stage_ms']['exciter_newt'], 'pipelined ms/step', p['ms_per_step'], '| realistic: one-stream', r['stage_ms']['exciter_newt'], 'ms/step', r['ms_per_step'])
PY
  done
done
