# kOptLowReg (80-register tail, three 8-wave workgroups per CU) against the default kernel: parity subset with the switch on, then
# same-box A/B of the one-stream kernel time and the pipelined step
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/lr
NWS_EXCITER_LOWREG=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "e2e or full_size or batch64 or hipgraph" > gpurun_out/lr/pytest_lowreg.txt 2>&1; tail -2 gpurun_out/lr/pytest_lowreg.txt
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0"
for i in 1 2; do
  for lr in 0 1; do
    NWS_EXCITER_LOWREG=$lr timeout 120 python bench.py $Q --pipeline 0 --streams 1 --steps 100 > gpurun_out/lr/one_$lr.json 2>/dev/null
    NWS_EXCITER_LOWREG=$lr timeout 120 python bench.py $Q > gpurun_out/lr/pipe_$lr.json 2>/dev/null
    NWS_EXCITER_LOWREG=$lr timeout 120 python bench.py $Q --inputs realistic > gpurun_out/lr/real_$lr.json 2>/dev/null
    python - <<PY
import json
o=json.loads(open('gpurun_out/lr/one_$lr.json').read().strip().splitlines()[-1]); p=json.loads(open('gpurun_out/lr/pipe_$lr.json').read().strip().splitlines()[-1]); r=json.loads(open('gpurun_out/lr/real_$lr.json').read().strip().splitlines()[-1])
print('lowreg', $lr, 'one-stream exciter ms', o['stage_ms']['exciter_newt'], 'pipelined ms/step', p['ms_per_step'], 'realistic ms/step', r['ms_per_step'])
PY
  done
done
