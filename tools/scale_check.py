#!/usr/bin/env python
"""First 8-GPU run in one call: bench.py at --gpus 1/2/4/8 (as far as the box has GPUs) x --gather rccl|copy x --gather-chunks 1|4,
one JSON line per cell under gpurun_out/scale/, then a table: samples/s, weak-scaling efficiency against the --gpus 1 line,
exchange.overlap_efficiency, step over the plain single-GPU pattern - and the winning exchange form per GPU count.

    python tools/scale_check.py [--steps 200] [--gpus 1,2,4,8] [--out gpurun_out/scale]
    python tools/scale_check.py --dry-run      # two ranks SHARING one GPU (gloo rendezvous; rccl cells = the collective branch on gloo), tiny shapes:
                                               # exercises every cell's code path where only one MI355X is available
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cell(n, gather, chunks, steps, out_dir, dry, extra, forced=False):
    name = f"g{n}_{gather}_c{chunks}" + ("_forced" if forced else "")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if forced:      # the N > 1 issue pattern at world size 1 (real RCCL, nothing to send): what the exchange machinery costs by itself
        env["NWS_BENCH_FORCE_DIST"] = "1"
    args = ["--gpus", str(n), "--steps", str(steps), "--warmup", "10", "--no-cpu-baseline", "--pmc", "off", "--legs", "0",
            "--batch1-iters", "0", "--gather", gather, "--gather-chunks", str(chunks), *extra]
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args]
    else:
        port = 29400 + (os.getpid() + 17 * n + 3 * chunks + (7 if gather == "copy" else 0)) % 500
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), *args]
        if dry:
            env["NWS_BENCH_SHARE_GPU"] = "1"
            env["NWS_BENCH_SHARE_GPU_COLLECTIVE"] = "1"      # --gather rccl: the collective branch, on gloo (RCCL refuses two ranks on one device)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    with open(os.path.join(out_dir, name + ".log"), "w") as f:
        f.write(r.stdout[-20000:] + "\n---- stderr ----\n" + r.stderr[-20000:])
    if r.returncode != 0 or not lines:
        return name, None
    with open(os.path.join(out_dir, name + ".json"), "w") as f:
        f.write(lines[-1] + "\n")
    return name, json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scale"))
    ap.add_argument("--dry-run", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    import torch
    have = torch.cuda.device_count()
    extra = []
    if a.dry_run:
        counts, steps = [1, 2], 4
        extra = ["--batch", "16", "--frames", "16"]
    else:
        counts, steps = [int(x) for x in a.gpus.split(",") if int(x) <= have], a.steps
        skipped = [int(x) for x in a.gpus.split(",") if int(x) > have]
        if skipped:
            print(f"scale_check: this box has {have} GPUs; skipping --gpus {skipped}")
    cells, base = [], None
    for n in counts:
        forms = [("rccl", 1)] if n == 1 else [(g, c) for g in ("rccl", "copy") for c in (1, 4)]
        for gather, chunks in forms:
            name, d = run_cell(n, gather, chunks, steps, a.out, a.dry_run, extra)
            cells.append((n, gather, chunks, name, d))
            if n == 1 and d is not None:
                base = d["value"]
        if n == 1 and not a.dry_run:
            for gather in ("rccl", "copy"):
                name, d = run_cell(1, gather, 1, steps, a.out, False, extra, forced=True)
                cells.append((1, gather, 1, name, d))
    print(f"{'cell':16s} {'ms/step':>9s} {'samples/s':>12s} {'efficiency':>10s} {'overlap':>8s} {'step/plain':>10s} {'rccl world':>10s} selfcheck")
    best = {}
    for n, gather, chunks, name, d in cells:
        if d is None:
            print(f"{name:16s}   FAILED (see {name}.log)")
            continue
        ex = d.get("exchange") or {}
        eff = d["value"] / (n * base) if base else float("nan")
        ratio = ex.get("step_over_single_gpu_pattern", ex.get("world1_overhead"))
        sc = d.get("pipeline_selfcheck") or {}
        print(f"{name:16s} {d['ms_per_step']:9.4f} {d['value']:12.4g} {eff:10.3f} {ex.get('overlap_efficiency', float('nan')):8.3f} "
              f"{ratio if ratio is not None else float('nan'):10.3f} {str(ex.get('rccl_world_size', '-')):>10s} "
              f"{sc.get('mismatching_all_ranks', sc.get('mismatching'))}")
        if n > 1 and (n not in best or d["value"] > best[n][1]):
            best[n] = (name, d["value"], eff)
    for n, (name, _, eff) in sorted(best.items()):
        print(f"winner at {n} GPUs: {name} (weak-scaling efficiency {eff:.3f})")
    failed = [c[3] for c in cells if c[4] is None]
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
