#!/bin/bash
# Same box, one call: the plain pipelined step for several histories of hardware queues, measured placement (default) against
# round 5's first-use order (NWS_PLACEMENT=order).  -> gpurun_out/placement_ab.txt (committed as profiles/r06/placement_ab.txt)
mkdir -p gpurun_out
out=gpurun_out/placement_ab.txt
echo "# plain single-GPU pipelined step (B = 64 x 4 s, 150 steps) after a history of hardware queues; tools/placement_ab.sh, one box, one call" > $out
echo "# history                      placement   ms/step   queue_offset  verified  ok   (probe = pipeline._place_by_measurement, order = round 5)" >> $out
for hist in "" "--pre 1" "--pre 2" "--pre 3" "--pre 3 --pre-high 2" "--pre 5" "--shape-first" "--pre 2 --shape-first" "--rccl-first"; do
  for mode in probe order; do
    NWS_PLACEMENT=$mode timeout 180 python tools/placement_case.py $hist 2>gpurun_out/placement_case.err | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    r=json.loads(l); p=r['placement']
    print(f\"{'$hist' or '(none)':30s} {'$mode':9s} {r['ms_per_step']:8.4f}   {str(p.get('queue_offset')):>6s}       {str(p.get('verified')):>5s}   {p['ok']}  {' '.join(p.get('blocked', []))}\")
except Exception as e:
    print('$hist $mode FAILED', e, l[:200])
" >> $out
  done
done
echo "# for scale: what a misplaced set costs on this box (round 5's NWS_STREAM_ORDER switch: d = a stream first used in between, as a second pipeline shape did then)" >> $out
for o in "x,a0,a1,c0,c1" "x,a0,a1,c0,d,c1" "x,a0,a1,c0,d,d,c1" "d,d,d,a0,a1,c0,c1,x" "a0,d,a1,c0,c1,x"; do
  NWS_STREAM_ORDER=$o timeout 180 python tools/placement_case.py 2>>gpurun_out/placement_case.err | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print(f\"first-use order {'$o':24s} {r['ms_per_step']:8.4f}\")" >> $out
done
cat $out
