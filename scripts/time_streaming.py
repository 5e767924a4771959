#!/usr/bin/env python
"""Latency of the STATEFUL streaming path (NewtStream.push): hops of `--hop-frames` control frames (2 = 256 samples =
16 ms of audio) pushed back to back, HIP-event timed per push.  Unlike scripts/time_buffer_sizes.py (stateless, hipGraph)
every push carries GRU / phase / noise / reverb state and applies the 2 s reverb as a linear overlap-add."""
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
def main(checkpoint, batch_size, hop_frames, num_hops, use_fast_newt):
    nws = importlib.import_module("neural-waveshaping-synthesis_amd")
    nws.ensure_default_config()
    model = nws.NeuralWaveshaping.load_from_checkpoint(checkpoint).cuda().eval()
    if use_fast_newt:
        model.newt = nws.FastNEWT(model.newt)
    K = hop_frames
    f0 = 220 + 20 * torch.rand(batch_size, 1, K, device="cuda")
    control = torch.randn(batch_size, 2, K, device="cuda")
    with torch.no_grad():
        s = model.stream(batch_size)
        for _ in range(20):
            s.push(f0, control)
        torch.cuda.synchronize()
        lat = []
        for _ in range(num_hops):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s.push(f0, control)
            e1.record()
            e1.synchronize()
            lat.append(e0.elapsed_time(e1) * 1e3)
    lat = np.array(lat)
    period = K * 128 / 16000.0 * 1e6
    print(f"stateful streaming, batch {batch_size}, hop {K * 128} samples ({period / 1e3:.1f} ms): p50 {np.percentile(lat, 50):.1f} us  "
          f"p99 {np.percentile(lat, 99):.1f} us  -> {period / np.percentile(lat, 50):.1f}x real-time")


if __name__ == "__main__":
    main()
