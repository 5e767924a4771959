#!/usr/bin/env python
"""First 8-GPU run in one call: bench.py at --gpus 1/2/4/8 (as far as the box has GPUs) x --gather rccl|copy x --gather-chunks 1|4,
one JSON line per cell under gpurun_out/scale/, then a table: samples/s, weak-scaling efficiency against the --gpus 1 line,
exchange.overlap_efficiency, step over the plain single-GPU pattern, the spread of the ranks' own ms/step, where the pipeline's
queues were placed - and the winning exchange form per GPU count.  Round 6 cells:
  * `rccl` with NCCL_MAX_NCHANNELS in {2, 4, 8} beside the default: how many CUs RCCL's ring kernels take from a VALU-saturated
    oscillator kernel (profiles/r06/cu_pressure.txt: 8-16 busy workgroups cost the step x1.04, 32 x1.05, 64 x1.08);
  * at the smallest N > 1: the measured placement after 1 / 2 / 3 streams used earlier in every rank (NWS_BENCH_PRE_STREAMS: the
    queue offset found must follow, the step must not), and round 5's first-use-order placement (NWS_PLACEMENT=order) for the A/B;
  * at one GPU: the 8-rank queue population rehearsed with seven local destinations (NWS_BENCH_FAKE_PEERS=7).
Every cell has a timeout and ONE retry; a failed cell prints the tail of its log and the run goes on.

    python tools/scale_check.py [--steps 200] [--gpus 1,2,4,8] [--out gpurun_out/scale] [--cell-timeout 600]
    python tools/scale_check.py --dry-run      # two ranks SHARING one GPU (gloo rendezvous; rccl cells = the collective branch on gloo), tiny shapes:
                                               # exercises every cell's code path where only one MI355X is available
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cell(n, gather, chunks, steps, out_dir, dry, extra, forced=False, tag="", env_extra=None, timeout=600, retries=1):
    name = f"g{n}_{gather}_c{chunks}" + ("_forced" if forced else "") + (f"_{tag}" if tag else "")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    if forced:      # the N > 1 issue pattern at world size 1 (real RCCL, nothing to send): what the exchange machinery costs by itself
        env["NWS_BENCH_FORCE_DIST"] = "1"
    args = ["--gpus", str(n), "--steps", str(steps), "--warmup", "10", "--no-cpu-baseline", "--pmc", "off", "--legs", "0",
            "--batch1-iters", "0", "--gather", gather, "--gather-chunks", str(chunks), *extra]
    if dry and n > 1:
        env["NWS_BENCH_SHARE_GPU"] = "1"
        env["NWS_BENCH_SHARE_GPU_COLLECTIVE"] = "1"      # --gather rccl: the collective branch, on gloo (RCCL refuses two ranks on one device)
    for attempt in range(retries + 1):
        if n == 1:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args]
        else:
            port = 29400 + (os.getpid() + 17 * n + 3 * chunks + (7 if gather == "copy" else 0) + 31 * attempt + sum(map(ord, tag))) % 500
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.join(ROOT, "bench.py"), *args]
        try:
            # a process group of its own: a hung cell is killed with every rank it started, by group id (never by pattern)
            p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                out, err = p.communicate(timeout=timeout)
                rc = p.returncode
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(p.pid, signal.SIGKILL)
                out, err = p.communicate()
                rc, err = -9, (err or "") + f"\nscale_check: cell timed out after {timeout} s"
        except OSError as e:
            out, err, rc = "", str(e), -1
        lines = [l for l in out.splitlines() if l.startswith("{")]
        with open(os.path.join(out_dir, name + (f".try{attempt}" if attempt else "") + ".log"), "w") as f:
            f.write(out[-20000:] + "\n---- stderr ----\n" + err[-20000:])
        if rc == 0 and lines:
            with open(os.path.join(out_dir, name + ".json"), "w") as f:
                f.write(lines[-1] + "\n")
            return name, json.loads(lines[-1])
        print(f"scale_check: cell {name} failed (rc {rc}, attempt {attempt + 1} of {retries + 1}); tail of its log:")
        print("    " + "\n    ".join((err or out).strip().splitlines()[-12:]))
    return name, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scale"))
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--cell-timeout", type=int, default=600, help="seconds per cell and attempt (a hung cell is killed by process group, retried once)")
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
    kw = dict(timeout=a.cell_timeout)
    multi = [n for n in counts if n > 1]
    for n in counts:
        forms = [("rccl", 1)] if n == 1 else [(g, c) for g in ("rccl", "copy") for c in (1, 4)]
        for gather, chunks in forms:
            name, d = run_cell(n, gather, chunks, steps, a.out, a.dry_run, extra, **kw)
            cells.append((n, gather, chunks, name, d))
            if n == 1 and d is not None:
                base = d["value"]
        if n > 1:
            # how many channels (= workgroups on the CUs) RCCL gets: fewer leave the oscillator kernel its vector issue, more move the
            # rows faster (on gloo, the dry run, the variable is inert: the cell only proves that the environment travels)
            for ch in ((2,) if a.dry_run else (2, 4, 8)):
                name, d = run_cell(n, "rccl", 1, steps, a.out, a.dry_run, extra, tag=f"ch{ch}", env_extra={"NCCL_MAX_NCHANNELS": str(ch)}, **kw)
                cells.append((n, "rccl", 1, name, d))
        if multi and n == multi[0]:
            # the measured placement must not care what the ranks created before their pipelines (the queue offset it reports
            # follows, the step does not); NWS_PLACEMENT=order: round 5's first-use order, for the A/B
            for pre in ((2,) if a.dry_run else (1, 2, 3)):
                name, d = run_cell(n, "rccl", 1, steps, a.out, a.dry_run, extra, tag=f"pre{pre}", env_extra={"NWS_BENCH_PRE_STREAMS": str(pre)}, **kw)
                cells.append((n, "rccl", 1, name, d))
            name, d = run_cell(n, "rccl", 1, steps, a.out, a.dry_run, extra, tag="order", env_extra={"NWS_PLACEMENT": "order"}, **kw)
            cells.append((n, "rccl", 1, name, d))
        if n == 1 and not a.dry_run:
            for gather in ("rccl", "copy"):
                name, d = run_cell(1, gather, 1, steps, a.out, False, extra, forced=True, **kw)
                cells.append((1, gather, 1, name, d))
            name, d = run_cell(1, "copy", 1, steps, a.out, False, extra, tag="fake7", env_extra={"NWS_BENCH_FAKE_PEERS": "7"}, **kw)
            cells.append((1, "copy", 1, name, d))
    def num(v, width):      # a single-GPU cell has no exchange: '-' rather than nan
        return f"{v:{width}.3f}" if isinstance(v, (int, float)) and v == v else f"{'-':>{width}s}"

    print(f"{'cell':22s} {'ms/step':>9s} {'samples/s':>12s} {'efficiency':>10s} {'overlap':>8s} {'step/plain':>10s} {'rank spread':>11s} {'rccl world':>10s} "
          f"{'placement':>16s} selfcheck")
    best = {}
    for n, gather, chunks, name, d in cells:
        if d is None:
            print(f"{name:22s}   FAILED (see {name}.log)")
            continue
        ex = d.get("exchange") or {}
        eff = d["value"] / (n * base) if base else float("nan")
        ratio = ex.get("step_over_single_gpu_pattern", ex.get("world1_overhead"))
        sc = d.get("pipeline_selfcheck") or {}
        pr = d.get("ms_per_step_per_rank") or [d["ms_per_step"]]
        pl = (d.get("config") or {}).get("placement") or {}
        where = f"{pl.get('mode', '-')}/{'ok' if pl.get('ok') else 'NOT OK'}/off{pl.get('queue_offset')}"
        print(f"{name:22s} {d['ms_per_step']:9.4f} {d['value']:12.4g} {eff:10.3f} {num(ex.get('overlap_efficiency'), 8)} "
              f"{num(ratio, 10)} {min(pr):5.3f}-{max(pr):5.3f} {str(ex.get('rccl_world_size', '-')):>10s} "
              f"{where:>16s} {sc.get('mismatching_all_ranks', sc.get('mismatching'))}")
        if n > 1 and (n not in best or d["value"] > best[n][1]):
            best[n] = (name, d["value"], eff)
    for n, (name, _, eff) in sorted(best.items()):
        print(f"winner at {n} GPUs: {name} (weak-scaling efficiency {eff:.3f})")
    failed = [c[3] for c in cells if c[4] is None]
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
