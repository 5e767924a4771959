#!/usr/bin/env python
"""MI355X counterpart of the reference's forward-pass timing CLI (same flags as its
scripts/time_forward_pass.py:13-22), but timed correctly for a GPU: HIP events on the launch stream,
warm-up iterations, explicit synchronisation.  Prints mean / p50 / p90 latency, the script's RTF convention
(time / audio duration, lower is better) and the x-real-time figure."""
import importlib
import os
import sys

import click
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@click.command()
@click.option("--gin-file", default=None, help="model gin file (default: the packaged newt.gin)")
@click.option("--num-iters", default=100)
@click.option("--batch-size", default=1)
@click.option("--device", default="cuda")
@click.option("--length-in-seconds", default=4)
@click.option("--sample-rate", default=16000)
@click.option("--control-hop", default=128)
@click.option("--use-fast-newt", is_flag=True)
@click.option("--checkpoint", default=None, help="optional .ckpt / .npz (default: random init, like the reference script)")
@click.option("--warmup", default=10)
def main(gin_file, num_iters, batch_size, device, length_in_seconds, sample_rate, control_hop, use_fast_newt, checkpoint,
         warmup):
    nws = importlib.import_module("neural-waveshaping-synthesis_amd")
    if gin_file:
        nws.gin.parse_config_file(gin_file)
    else:
        nws.ensure_default_config()
    T = sample_rate * length_in_seconds // control_hop
    dummy_control = torch.rand(batch_size, 2, T, device=device)
    dummy_f0 = torch.rand(batch_size, 1, T, device=device)
    model = nws.NeuralWaveshaping.load_from_checkpoint(checkpoint) if checkpoint else nws.NeuralWaveshaping()
    if use_fast_newt:
        model.newt = nws.FastNEWT(model.newt)
    model = model.eval().to(device)
    times = []
    with torch.no_grad():
        for _ in range(warmup):
            model(dummy_f0, dummy_control)
        torch.cuda.synchronize()
        for _ in range(num_iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model(dummy_f0, dummy_control)
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e-3)
    t = np.array(times)
    print(f"iters {len(t)}  mean {t.mean() * 1e3:.4f} ms  p50 {np.percentile(t, 50) * 1e3:.4f} ms  "
          f"p90 {np.percentile(t, 90) * 1e3:.4f} ms  min {t.min() * 1e3:.4f} ms")
    rtfs = t / length_in_seconds
    print("Mean RTF: %.6f" % np.mean(rtfs))
    print("90th percentile RTF: %.6f" % np.percentile(rtfs, 90))
    print("x real-time (batch aggregate): %.1f" % (batch_size * length_in_seconds / t.mean()))
    print("samples/s: %.4e" % (batch_size * T * control_hop / t.mean()))


if __name__ == "__main__":
    main()
