# occupancy sensitivity of the hot oscillator kernel: unused LDS per workgroup -> 2 / 1 workgroups per CU (4 / 2 waves per SIMD)
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/occ
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --pipeline 0 --streams 1 --steps 100"
for pad in 0 45000 0 45000; do
  NWS_EXCITER_LDS_PAD=$pad timeout 120 python bench.py $Q > gpurun_out/occ/pad_$pad.json 2>/dev/null
  python -c "
import json,sys
d=json.loads(open('gpurun_out/occ/pad_$pad.json').read().strip().splitlines()[-1]); print('pad',$pad,d['ms_per_step'],d['stage_ms'])"
done
