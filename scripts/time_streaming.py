#!/usr/bin/env python
"""Latency of the STATEFUL streaming path (NewtStream.push): hops of `--hop-frames` control frames (2 = 256 samples =
16 ms of audio) pushed back to back, HIP-event timed per push.  Unlike scripts/time_buffer_sizes.py (stateless, hipGraph)
every push carries GRU / phase / noise / reverb state and applies the 2 s reverb as a linear convolution of the stream
(csrc/stream.hip); steady-state hops replay a hipGraph unless --no-graph.  Host wall-clock per push is reported beside the
HIP-event latency (a push returns a fresh tensor: input copy + graph launch + output copy)."""
import importlib
import os
import sys

import click
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@click.command()
@click.option("--checkpoint", default=os.path.join(ROOT, "tests", "golden", "weights_vn.npz"))
@click.option("--batch-size", default=1)
@click.option("--hop-frames", default=2)
@click.option("--num-hops", default=500)
@click.option("--use-fast-newt/--no-fast-newt", default=True)
@click.option("--graph/--no-graph", default=True)
@click.option("--static-io/--copy-io", default=False, help="time NewtStream.hop(): the caller fills the captured hop's own "
              "input buffers and reads its output buffer (no input / output copies), like an audio callback would")
@click.option("--json-out", default=None)
@click.option("--gc/--no-gc", "keep_gc", default=True, help="--no-gc: Python's cyclic garbage collector off during the timed hops "
              "(what a host with a real-time audio callback does); reported in the result")
def main(checkpoint, batch_size, hop_frames, num_hops, use_fast_newt, graph, static_io, json_out, keep_gc):
    nws = importlib.import_module("neural-waveshaping-synthesis_amd")
    nws.ensure_default_config()
    model = nws.NeuralWaveshaping.load_from_checkpoint(checkpoint).cuda().eval()
    if use_fast_newt:
        model.newt = nws.FastNEWT(model.newt)
    K = hop_frames
    f0 = 220 + 20 * torch.rand(batch_size, 1, K, device="cuda")
    control = torch.randn(batch_size, 2, K, device="cuda")
    with torch.no_grad():
        s = model.stream(batch_size, graph=graph)
        for _ in range(20):
            s.push(f0, control)
        if static_io:
            f0_in, c_in, _ = s.static_io(K)
            f0_in.copy_(f0[:, 0])
            c_in.copy_(control)
        torch.cuda.synchronize()
        import gc
        import time
        lat, wall = [], []
        if not keep_gc:
            gc.collect()
            gc.disable()
        for _ in range(num_hops):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            if static_io:
                s.hop(K)
            else:
                s.push(f0, control)
            e1.record()
            e1.synchronize()
            wall.append((time.perf_counter() - t0) * 1e6)
            lat.append(e0.elapsed_time(e1) * 1e3)
    gc.enable()
    lat, wall = np.array(lat), np.array(wall)
    period = K * 128 / 16000.0 * 1e6
    res = {"batch": batch_size, "hop_samples": K * 128, "graph": bool(graph), "static_io": bool(static_io), "hops": num_hops, "python_gc": bool(keep_gc),
           "p50_us": float(np.percentile(lat, 50)), "p99_us": float(np.percentile(lat, 99)), "max_us": float(lat.max()),
           "wall_p50_us": float(np.percentile(wall, 50)), "wall_p99_us": float(np.percentile(wall, 99)),
           "x_realtime_p50": period / float(np.percentile(lat, 50))}
    print(f"stateful streaming, batch {batch_size}, hop {K * 128} samples ({period / 1e3:.1f} ms), graph={graph}, static_io={static_io}, gc={keep_gc}: p50 {res['p50_us']:.1f} us  "
          f"p99 {res['p99_us']:.1f} us  (host wall p50 {res['wall_p50_us']:.1f} / p99 {res['wall_p99_us']:.1f} us)  -> "
          f"{res['x_realtime_p50']:.1f}x real-time")
    if json_out:
        import json
        with open(json_out, "a") as f:
            f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
