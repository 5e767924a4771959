# Where HIP's hardware queues land on the command processor's pipes: the pipelined step with the pipeline's streams FIRST USED in
# different orders (NWS_STREAM_ORDER of pipeline.placed_streams: x exchange, a0 a1 audio, c0 c1 control, d / h a dummy normal- /
# high-priority stream never used again).  Part 1: the single-GPU step (the four critical streams need four different pipes:
# period 4 in the number of queues created in between).  Part 2: where the exchange queue may sit (N > 1 issue pattern at world size
# 1 with three tiny launches per step on it, NWS_BENCH_DIAG=blit3).  profiles/r05/queue_placement.txt is a run of this script.
export TMPDIR=/tmp
rm -rf gpurun_out/qo; mkdir -p gpurun_out/qo
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps 150 --warmup 10"
ORDERS=${ORDERS:-"a0,a1,c0,c1 a0,a1,c0,d,c1 a0,a1,c0,d,d,c1 a0,a1,c0,d,d,d,c1 a0,a1,c0,d,d,d,d,c1 a0,a1,c0,h,c1 a0,a1,c0,h,h,c1 a0,d,a1,c0,c1 c0,c1,d,a0,a1 x,a0,a1,c0,c1"}
i=0
for o in $ORDERS; do
  i=$((i+1)); n=$(printf "%02d" $i)
  NWS_STREAM_ORDER=$o timeout 120 python bench.py $Q > gpurun_out/qo/${n}_single_${o//,/-}.json 2>/dev/null
done
XORDERS=${XORDERS:-"x,a0,a1,c0,c1 a0,a1,c0,c1,x a0,a1,c0,c1,d,x a0,a1,c0,c1,d,d,x a0,a1,c0,c1,d,d,d,x"}
for o in $XORDERS; do
  i=$((i+1)); n=$(printf "%02d" $i)
  NWS_STREAM_ORDER=$o NWS_BENCH_DIAG=blit3 NWS_BENCH_FORCE_DIST=1 timeout 120 python tools/world1_diag.py $Q --gather rccl > gpurun_out/qo/${n}_blit3_${o//,/-}.json 2>/dev/null
done
python - <<'PY'
import json, glob, os
for p in sorted(glob.glob("gpurun_out/qo/*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1]); ex = d.get("exchange") or {}
        extra = f"  world1_overhead {ex['world1_overhead']:.4f}  plain {ex['single_gpu_pattern_ms']:.4f}" if ex else ""
        print(f"{os.path.basename(p)[3:-5].replace('-', ','):44s} {d['ms_per_step']:.4f}{extra}")
    except Exception as e:
        print(p, "failed", e)
PY
