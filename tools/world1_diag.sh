# What the N > 1 issue pattern at world size 1 is made of (NWS_BENCH_DIAG switches of tools/world1_diag.py = bench.py's main() with diagnosis hooks), same box:
#   (none)   the product pattern: completion-driven exchange on the placed exchange stream
#   noexch   the mechanism alone (events, helper thread), nothing issued
#   blit3    three tiny launches per step on the exchange stream in place of the collective
#   queued   the round-4 form: the exchange enqueued behind the batch by a device-side wait on the exchange queue (+24-30 %)
# wprof adds the helper thread's per-exchange times (wait / issue / record) to the line.
export TMPDIR=/tmp
rm -rf gpurun_out/w1d; mkdir -p gpurun_out/w1d
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps 200"
timeout 120 python bench.py $Q > gpurun_out/w1d/00_single.json 2>/dev/null
i=0
for g in rccl copy; do
  for d in wprof wprof,noexch wprof,blit3 queued; do
    i=$((i+1)); n=$(printf "%02d" $i)_${g}_${d//,/_}
    NWS_BENCH_DIAG=$d NWS_BENCH_FORCE_DIST=1 timeout 120 python tools/world1_diag.py $Q --gather $g > gpurun_out/w1d/$n.json 2> gpurun_out/w1d/$n.err
  done
done
python - <<'PY'
import json, glob, os
for p in sorted(glob.glob("gpurun_out/w1d/*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1]); ex = d.get("exchange") or {}
        w1 = f"{ex['world1_overhead']:.4f}" if ex else "-"
        print(f"{os.path.basename(p)[:-5]:32s} {d['ms_per_step']:.4f}  world1_overhead {w1}  host {d.get('host_issue_ms_per_step')}  worker {d.get('exchange_worker_us')}")
    except Exception as e:
        print(p, "failed", e)
PY
