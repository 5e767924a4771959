#!/bin/bash
# Rehearsal of the 8-rank queue population on ONE GPU (VERDICT r5 #1b), same box, one call: the N > 1 issue pattern at forced
# world size 1 with `--gather copy` pushing every step's 16.4 MB shard to k LOCAL buffers on k per-peer copy streams
# (NWS_BENCH_FAKE_PEERS=k; blit kernels stand in for the copy engines), beside the single-GPU line and the real RCCL / copy
# forms with nothing to send.  -> gpurun_out/fake_peers_ab.txt (committed as profiles/r06/fake_peers_ab.txt)
export TMPDIR=/tmp
mkdir -p gpurun_out/fp
Q="--no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --steps ${K:-200}"
run() {   # name, env..., -- args
    name=$1; shift
    envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout 200 python bench.py $Q "$@" > gpurun_out/fp/$name.json 2> gpurun_out/fp/$name.err
}
run 00_single X=1 --
run 01_rccl NWS_BENCH_FORCE_DIST=1 -- --gather rccl
run 02_copy NWS_BENCH_FORCE_DIST=1 -- --gather copy
run 03_fake1 NWS_BENCH_FAKE_PEERS=1 --
run 04_fake3 NWS_BENCH_FAKE_PEERS=3 --
run 05_fake7 NWS_BENCH_FAKE_PEERS=7 --
run 06_fake7_plain_streams NWS_BENCH_FAKE_PEERS=7 NWS_BENCH_COPY_STREAMS=plain --
run 07_fake7_event_per_copy NWS_BENCH_FAKE_PEERS=7 NWS_PEER_EVENTS=1 --
run 08_fake7_1row NWS_BENCH_FAKE_PEERS=7 NWS_BENCH_FAKE_ROWS=1 --
run 09_fake7_24_hw_queues NWS_BENCH_FAKE_PEERS=7 GPU_MAX_HW_QUEUES=24 --
run 10_fake7_chunks4 NWS_BENCH_FAKE_PEERS=7 -- --gather-chunks 4
run 11_single_b X=1 --
for extra in "$@"; do eval "$extra"; done
python - <<'PY' | tee gpurun_out/fake_peers_ab.txt
import json, glob, os
print("# N > 1 issue pattern at forced world size 1, B = 64 x 4 s per step; fakeK = --gather copy pushing the 16.4 MB shard to K local buffers on K copy streams (tools/fake_peers_ab.sh)")
base = None
for p in sorted(glob.glob("gpurun_out/fp/*.json")):
    n = os.path.basename(p)[:-5]
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "failed", e, open(p[:-5] + ".err").read()[-600:]); continue
    if base is None:
        base = d["ms_per_step"]
    ex = d.get("exchange") or {}
    f = lambda v: "-" if v is None else f"{v:.4f}"
    pl = (d.get("config") or {}).get("placement") or {}
    print(f"{n:18s} ms/step {d['ms_per_step']:.4f}  x{d['ms_per_step']/base:.3f} of the first line  host_issue {d.get('host_issue_ms_per_step')}  "
          f"world1_overhead {f(ex.get('world1_overhead'))}  single_pattern {f(ex.get('single_gpu_pattern_ms'))}  compute_only {f(ex.get('compute_only_ms'))}  "
          f"gather {f(ex.get('gather_ms'))}  overlap_eff {f(ex.get('overlap_efficiency'))}  fake_peers {ex.get('fake_peers', '-')}  "
          f"placement ok={pl.get('ok')} verified={pl.get('verified')} offset={pl.get('queue_offset')} side={pl.get('side')}  "
          f"selfcheck {(d.get('pipeline_selfcheck') or {}).get('mismatching_all_ranks', (d.get('pipeline_selfcheck') or {}).get('mismatching'))}")
PY
