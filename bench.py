#!/usr/bin/env python
"""Headline benchmark: NEWT forward throughput (audio samples/s, x real-time) on 4 s @ 16 kHz clips.

    python bench.py [--gpus N --steps K --warmup W] [--batch 64] [--frames 500] [--exact]

Contract: a step = one NeuralWaveshaping.forward over a batch of `--batch` synthetic
utterances per GPU (torch.rand F0/control exactly like the reference's scripts/time_forward_pass.py:27-40,
vn checkpoint, FastNEWT LUT), inputs resident in HBM, the two RNG draws of forward() made on the device
inside the step; with N>1 the rendered waveforms are all-gathered over RCCL (overlapped with the next
step's kernels).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

# the pipeline keeps 2 audio + 1 control stream busy next to torch's default stream and, for N > 1, RCCL's own streams; HIP maps
# streams onto 4 hardware queues by default and streams that share a queue serialise (measured: 0.75 vs 0.59 ms per step with
# RCCL initialised).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work of the dominant kernel (exciter_newt_kernel) per utterance of T=500 frames, SURVEY.md §8(d):
#   harmonic mixer 2*64*101 flop/sample + 101 sin/sample (1 flop each) + FiLM lerp/FiLM/LUT/mix ~ 24 flop per (sample, shaper)
FLOP_PER_SAMPLE_EXCITER_NEWT = 2 * 64 * 101 + 101 * 5 + 64 * 24
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= the fp32 vector peak)
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
# matrix-core work the kernel actually issues: 3 fp16 MFMAs (hi*hi, hi*lo, lo*hi) of 64 shapers x 112 K slots per sample
MFMA_F16_FLOP_PER_SAMPLE_EXECUTED = 3 * 2 * 64 * 112


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU (weak scaling: fixed per-GPU work)")
    ap.add_argument("--frames", type=int, default=500, help="control frames per utterance (500 = 4 s @ 16 kHz)")
    ap.add_argument("--exact", action="store_true", help="exact sin-MLP shapers instead of the FastNEWT LUT")
    ap.add_argument("--inputs", choices=("rand", "realistic"), default="rand",
                    help="rand: torch.rand F0/control exactly like scripts/time_forward_pass.py (sub-1 Hz 'F0': all 101 harmonics "
                         "live, the worst case for the oscillator); realistic: per-utterance F0 ~ U[100, 1000] Hz with 5.5 Hz "
                         "vibrato, control ~ N(0,1) (SURVEY 8(d) config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=60)
    ap.add_argument("--batch1-iters", type=int, default=200)
    ap.add_argument("--streams", type=int, default=2,
                    help="audio streams of the pipeline (or, with --pipeline 0, streams that whole forwards are issued on "
                         "round-robin).  Two overlap the tail of batch i with the head of batch i+1 (~7 %%); this is the "
                         "configuration that exposed the packed-fp32 / MFMA co-execution hazard the build now guards against "
                         "(DESIGN.md 5.2) - the self-check below compares it with the plain forward bit for bit")
    ap.add_argument("--depth", type=int, default=0, help="workspaces in flight (0: streams + 2)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1: ForwardPipeline (control half = carries + GRU of batch i+1 on a side stream under the audio "
                         "half of batch i, --streams audio streams); 0: whole forwards round-robin on --streams streams")
    ap.add_argument("--control-streams", type=int, default=1,
                    help="side streams for the control half (two change nothing for rand inputs, help the GRU-bound realistic-"
                         "input case by 7 %% (0.392 -> 0.364 ms/step) and hurt beside RCCL: 0.62 vs 0.48 ms/step with the "
                         "all-gather in the loop)")
    ap.add_argument("--gru", choices=("batched", "per-utterance"), default="per-utterance",
                    help="GRU kernel of the pipeline's control half (the plain forward always uses per-utterance)")
    return ap.parse_args()


def cpu_baseline(weights_path, iters, T):
    """The oracle (op-for-op torch-CPU restatement of the reference forward, python LUT loop included) timed on
    the host cores, protocol of scripts/time_forward_pass.py: B=1, torch.rand inputs, FastNEWT."""
    from oracle.newt_oracle import OracleNEWT, load_weights_npz

    w = {k: v for k, v in load_weights_npz(weights_path).items() if not k.startswith("__")}
    o = OracleNEWT(w, fast=True, lut_python_loop=True)
    torch.manual_seed(0)
    f0, control = torch.rand(1, 1, T), torch.rand(1, 2, T)
    for _ in range(3):
        o(f0, control)
    ts = []
    t_end = time.time() + 25.0
    for _ in range(iters):
        t0 = time.time()
        o(f0, control)
        ts.append(time.time() - t0)
        if time.time() > t_end:
            break
    mean = float(np.mean(ts))
    return {"value": 128 * T / mean, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(ts)} forwards of B=1, T={T} (4 s) FastNEWT, torch.rand inputs, oracle/newt_oracle.py "
                      f"(torch {torch.__version__} CPU, {os.cpu_count()} logical cpus)",
            "ms_per_utterance": mean * 1e3, "x_realtime": (128 * T / 16000.0) / mean}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback for the product path)"
    # NWS_BENCH_SHARE_GPU=1 (smoke-testing the N>1 code path on a 1-GPU box): every rank uses cuda:0 and the
    # collectives go through gloo on host copies (NCCL/RCCL refuses two ranks on one device).  Never set by the driver.
    share_gpu = os.environ.get("NWS_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NWS_BENCH_FORCE_DIST=1: run the N>1 code path (RCCL init, shared draws, all-gather) with world_size 1 -- the only way
    # to exercise it on real RCCL on a 1-GPU box.  Never set by the driver.
    distributed = world > 1 or os.environ.get("NWS_BENCH_FORCE_DIST") == "1"
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    nws = importlib.import_module("neural-waveshaping-synthesis_amd")
    _lib = importlib.import_module("neural-waveshaping-synthesis_amd._lib")
    par = importlib.import_module("neural-waveshaping-synthesis_amd.parallel")
    nws.ensure_default_config()
    wpath = os.path.join(ROOT, "tests", "golden", "weights_vn.npz")
    model = nws.NeuralWaveshaping.load_from_checkpoint(wpath).to(dev).eval()
    if not a.exact:
        model.newt = nws.FastNEWT(model.newt)

    B, T = a.batch, a.frames
    N = 128 * T
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    if a.inputs == "rand":
        f0 = torch.rand(B, 1, T, device=dev, generator=g)         # time_forward_pass.py:34-40
        control = torch.rand(B, 2, T, device=dev, generator=g)    # time_forward_pass.py:27-33
    else:
        tt = torch.arange(T, device=dev, dtype=torch.float32) * (128.0 / 16000.0)
        base = 100.0 + 900.0 * torch.rand(B, 1, 1, device=dev, generator=g)
        f0 = (base * (1.0 + 0.01 * torch.sin(2 * np.pi * 5.5 * tt).view(1, 1, T))).contiguous()
        control = torch.randn(B, 2, T, device=dev, generator=g)

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, a.streams))]
    shared_gen = par.make_shared_generator(dev) if distributed else None   # same draws on every rank, no broadcast
    use_pipe = bool(a.pipeline) and not share_gpu
    pipe = None
    if use_pipe:
        pmod = importlib.import_module("neural-waveshaping-synthesis_amd.pipeline")
        pipe = pmod.ForwardPipeline(model, depth=a.depth if a.depth > 0 else len(streams) + 2,
                                    audio_streams=len(streams), control_streams=max(1, a.control_streams), batched_gru=a.gru == "batched")
        streams = pipe.audio
    nbuf = len(pipe.slots) if use_pipe else len(streams)
    full = [torch.empty((B * world, N), dtype=torch.float32, device=dev) for _ in range(nbuf)] if distributed else None

    def step(i, pending):
        if use_pipe:
            if not distributed:
                pipe.submit(f0, control)
                return None
            au = pipe.audio[i % len(pipe.audio)]        # the audio stream submit() is about to use for this batch
            with torch.cuda.stream(au):                 # draws where they are consumed: nothing ever runs on the null stream
                pu, nz = par.shared_draws(101, N - 1, dev, generator=shared_gen)   # identical on all ranks
            y = pipe.submit(f0, control, phase_u=pu, noise=nz)
            with torch.cuda.stream(au):                 # ordered after this batch's reverb
                if pending is not None:
                    pending.wait()
                return dist.all_gather_into_tensor(full[i % nbuf], y, async_op=True)
        s = streams[i % len(streams)]
        with torch.cuda.stream(s):
            if distributed and share_gpu:   # smoke mode only: host-staged gloo collectives
                pu, nz = par.shared_draws(101, N - 1, torch.device("cpu"))
                y = model(f0, control, phase_u=pu.to(dev), noise=nz.to(dev))
                parts = [torch.empty((B, N)) for _ in range(world)]
                dist.all_gather(parts, y.cpu())
                full[i % nbuf].copy_(torch.cat(parts, 0))
                return None
            if distributed:
                pu, nz = par.shared_draws(101, N - 1, dev, generator=shared_gen)   # identical on all ranks
                y = model(f0, control, phase_u=pu, noise=nz)
                if pending is not None:
                    pending.wait()
                return dist.all_gather_into_tensor(full[i % nbuf], y, async_op=True)
            model(f0, control)
        return None

    def join_streams():
        if use_pipe:
            pipe.synchronize()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)

    # Priming, before the W warmup steps and untimed like them (reported as priming_steps): (1) every stream's caching-
    # allocator pool and every workspace of the ring has to have been through one full cycle, or the first 2 * depth timed
    # steps contain hipMallocs (measured: 1.2 ms/step instead of 0.47 for K = 50, W = 5 with 6 workspaces in flight);
    # (2) ~50 ms of continuous load before the clock starts helps short runs a little (K = 50: 0.504 -> 0.491 ms/step).
    # What stays inside the timed region by construction is the pipeline's fill and drain (first control half, last audio
    # half: 0.5-1 ms per timed region): 0.518 ms/step at K = 10, 0.491 at 50, 0.471 at 200, 0.468 at 1000.
    priming = max(2 * nbuf + 2, 120)
    with torch.no_grad():
        pending = None
        for i in range(priming + a.warmup):
            pending = step(i, pending)
        if pending is not None:
            pending.wait()
        join_streams()
        torch.cuda.synchronize()
        # live HIP-event timing of the dominant kernel on its launch stream, inside the timed region
        _lib.check(_lib.lib().nws_profile_begin(a.steps, (1 << 3) | (1 << 1)))
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending = None
        for i in range(a.steps):
            pending = step(i, pending)
        if pending is not None:
            pending.wait()
        join_streams()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        ms = (C.c_float * (a.steps * 6))()
        n = C.c_int(0)
        _lib.check(_lib.lib().nws_profile_collect(ms, C.byref(n)))
        k_ms = float(np.mean([ms[i * 6 + 3] for i in range(n.value)])) if n.value else float("nan")
        gru_ms_live = float(np.mean([ms[i * 6 + 1] for i in range(n.value)])) if n.value else float("nan")

        if distributed:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            # the gathered batch must hold every rank's rows (rank r at rows [r*B, (r+1)*B))
            last = full[(a.steps - 1) % nbuf]
            assert torch.isfinite(last).all() and float(last[B * (world - 1):].abs().max()) > 0.0

        extra = {}
        if use_pipe:
            # self-check outside the timed region: the issue pattern of the timed loop (pipeline, and the all-gather when
            # distributed) must return bit for bit what a plain forward returns for the same draws
            _lib.lib().nws_profile_end()
            gchk = torch.Generator(device=dev).manual_seed(4242 + rank)
            draws = [(torch.rand(101, device=dev, generator=gchk), torch.rand(N - 1, device=dev, generator=gchk))
                     for _ in range(12)]
            torch.cuda.synchronize()
            ys, pend = [], None
            for i, (pu_c, nz_c) in enumerate(draws):
                y = pipe.submit(f0, control, phase_u=pu_c, noise=nz_c)
                ys.append(y)
                if distributed:
                    with torch.cuda.stream(pipe.audio[i % len(pipe.audio)]):
                        if pend is not None:
                            pend.wait()
                        pend = dist.all_gather_into_tensor(full[i % nbuf], y, async_op=True)
            if pend is not None:
                pend.wait()
            pipe.synchronize()
            torch.cuda.synchronize()
            wrong = 0
            for y, (pu_c, nz_c) in zip(ys, draws):
                wrong += 0 if torch.equal(y, model(f0, control, phase_u=pu_c, noise=nz_c)) else 1
            torch.cuda.synchronize()
            extra["pipeline_selfcheck"] = {"batches": len(ys), "mismatching": wrong}
            if wrong:   # reported, not raised: the line must still come out, with the evidence in it
                print(f"bench.py: pipeline self-check FAILED on rank {rank}: {wrong} of {len(ys)} batches differ from the plain "
                      f"forward", file=sys.stderr)
            if distributed:   # every rank's verdict
                tw = torch.tensor([wrong], dtype=torch.float64, device="cpu" if share_gpu else dev)
                dist.all_reduce(tw, op=dist.ReduceOp.SUM)
                extra["pipeline_selfcheck"]["mismatching_all_ranks"] = int(tw.item())
        if rank == 0:
            # per-stage breakdown (diagnostic, outside the timed region)
            _lib.check(_lib.lib().nws_profile_begin(10, 0x3F))
            for _ in range(10):
                model(f0, control)
            torch.cuda.synchronize()
            ms2 = (C.c_float * 60)()
            _lib.check(_lib.lib().nws_profile_collect(ms2, C.byref(n)))
            extra["stage_ms"] = {nm: round(float(np.mean([ms2[i * 6 + s] for i in range(n.value)])), 4)
                                 for s, nm in enumerate(_lib.STAGE_NAMES)}   # one stream, nothing overlapping
            _lib.lib().nws_profile_end()
            if world == 1 and a.batch1_iters > 0:
                # config "batch=1, single MI355X, FastNEWT": latency / x real-time per utterance
                f1, c1 = f0[:1].contiguous(), control[:1].contiguous()
                for _ in range(10):
                    model(f1, c1)
                torch.cuda.synchronize()
                lat = []
                for _ in range(a.batch1_iters):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    model(f1, c1)
                    e1.record()
                    e1.synchronize()
                    lat.append(e0.elapsed_time(e1))
                lat = np.array(lat)
                dur_ms = N / 16000.0 * 1e3
                extra["batch1"] = {"p50_ms": round(float(np.percentile(lat, 50)), 4), "p90_ms": round(float(np.percentile(lat, 90)), 4),
                                   "x_realtime_p50": round(dur_ms / float(np.percentile(lat, 50)), 1),
                                   "rtf_mean": float(np.mean(lat) / dur_ms)}

    if rank == 0 and world == 1 and a.batch1_iters > 0:
        # config "streaming 256-sample hop": stateless forward of one 16 ms buffer, captured once into a hipGraph
        with torch.no_grad():
            fs, cs = torch.rand(1, 1, 2, device=dev), torch.rand(1, 2, 2, device=dev)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    model(fs, cs)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                model(fs, cs)
            for _ in range(20):
                graph.replay()
            torch.cuda.synchronize()
            lat = []
            for _ in range(500):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graph.replay()
                e1.record()
                e1.synchronize()
                lat.append(e0.elapsed_time(e1) * 1e3)
            extra["streaming_hop256_hipgraph"] = {"p50_us": round(float(np.percentile(lat, 50)), 2),
                                                  "p99_us": round(float(np.percentile(lat, 99)), 2),
                                                  "buffer_period_us": 16000.0}
    if rank == 0:
        total_samples = B * world * N * a.steps
        value = total_samples / elapsed
        ms_per_step = elapsed / a.steps * 1e3
        flops = FLOP_PER_SAMPLE_EXCITER_NEWT * B * N
        achieved = flops / (k_ms * 1e-3) / 1e12
        traffic, valu_issue = None, None   # from the committed rocprofv3 PMC passes (profiles/r01/pmc_traffic.json)
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))
            if tr["batch_per_gpu"] == B and tr["frames"] == T and not a.exact:
                traffic = tr["hbm_bytes_per_launch"]
                if tr.get("valu_busy_frac"):
                    # VALU-busy share of the kernel: (waves x VALU-active cycles per wave) / (1024 SIMDs x kernel cycles)
                    valu_issue = {"insts_per_wave": round(tr["valu_insts_per_wave"], 1),
                                  "trans_per_wave": round(tr["trans_insts_per_wave"], 1),
                                  "busy_frac_of_kernel_cycles": round(tr["valu_busy_frac"], 3),
                                  "source": "rocprofv3 --pmc, one stream (profiles/r01/pmc_traffic.json)"}
        except Exception:
            pass
        mfma_exec = MFMA_F16_FLOP_PER_SAMPLE_EXECUTED * B * N / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "audio_samples_per_sec", "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "priming_steps": priming, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"NEWT forward, vn checkpoint, {'exact sin-MLP shapers' if a.exact else 'FastNEWT LUT'}, "
                                   f"batch {B}/GPU x {T} frames (4 s @ 16 kHz), "
                                   f"{'torch.rand F0/control' if a.inputs == 'rand' else 'F0 ~ U[100,1000] Hz with vibrato, control ~ N(0,1)'}, "
                                   f"RNG draws on device{', RCCL all-gather of waveforms' if distributed else ''}",
                       "batch_per_gpu": B, "frames": T, "samples_per_utterance": N, "parallelism": f"batch-shard x{world}",
                       "streams": len(streams),
                       "issue": (f"ForwardPipeline: control half (carries + {a.gru} GRU) on {len(pipe.control)} side stream(s), "
                                 f"audio half on {len(streams)} streams, {len(pipe.slots)} workspaces in flight") if use_pipe
                       else f"whole forwards round-robin on {len(streams)} streams"},
            "x_realtime_aggregate": value / 16000.0,
            "rtf_per_utterance": (ms_per_step * 1e-3) / (N / 16000.0) / B,
            "roofline": {"bound": "mfma", "kernel": "exciter_newt_kernel", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                         "kernel_ms": k_ms, "flop_per_launch": flops, "gru_ms_in_timed_region": gru_ms_live,
                         "note": "achieved = algorithmic fp32 flop per launch / live kernel time, peak = fp32 matrix (= vector) "
                                 "peak as the path computes in f32.  The 101->64 contraction runs as three fp16 MFMAs per fp32 "
                                 "product on the fp16 matrix pipe (16x the fp32 MFMA rate), which is how frac can pass 1; the "
                                 "kernel is bound by VALU issue (sines, table index math), see valu_issue and DESIGN.md 3.2",
                         "mfma_f16_executed": {"achieved": mfma_exec, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                               "frac": mfma_exec / PEAK_F16_MFMA_TFLOPS},
                         "valu_issue": valu_issue,
                         # the same kernel with nothing else in flight (diagnostic pass, one stream): steps of the timed
                         # region overlap on --streams HIP streams, which stretches each individual launch
                         "kernel_ms_isolated": extra.get("stage_ms", {}).get("exciter_newt"),
                         "frac_isolated": (flops / (extra["stage_ms"]["exciter_newt"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS)
                         if extra.get("stage_ms") else None},
        }
        out.update(extra)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wpath, a.cpu_iters, T)
    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        try:   # RCCL prints its banner through C stdio: drain that first so that the JSON line is the last line of stdout
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
