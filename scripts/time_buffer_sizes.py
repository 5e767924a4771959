#!/usr/bin/env python
"""MI355X counterpart of the reference's streaming-buffer benchmark (scripts/time_buffer_sizes.py: buffer sizes
256 ... 32768 samples, stateless forward per buffer).  Each buffer size is captured ONCE into a hipGraph
(torch.cuda.CUDAGraph around the single nws_forward enqueue, RNG draws included) and replayed; latency is
measured per replay with HIP events.  Writes the reference's CSV row format [model, "gpu", buffer_size, seconds]
and prints p50 / p99 per size next to the buffer period."""
import csv
import importlib
import os
import sys

import click
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUFFER_SIZES = [256, 512, 1024, 2048, 4096, 8192, 16384, 32768]


@click.command()
@click.option("--gin-file", default=None)
@click.option("--output-file", default=None, help="CSV path (reference row format)")
@click.option("--num-iters", default=1000)
@click.option("--batch-size", default=1)
@click.option("--device", default="cuda")
@click.option("--use-fast-newt", is_flag=True)
@click.option("--model-name", default="ours-mi355x")
@click.option("--checkpoint", default=None)
@click.option("--no-graph", is_flag=True, help="eager launches instead of hipGraph replay")
def main(gin_file, output_file, num_iters, batch_size, device, use_fast_newt, model_name, checkpoint, no_graph):
    nws = importlib.import_module("neural-waveshaping-synthesis_amd")
    if gin_file:
        nws.gin.parse_config_file(gin_file)
    else:
        nws.ensure_default_config()
    model = nws.NeuralWaveshaping.load_from_checkpoint(checkpoint) if checkpoint else nws.NeuralWaveshaping()
    if use_fast_newt:
        model.newt = nws.FastNEWT(model.newt)
    model = model.eval().to(device)
    rows, summary = [], {}
    with torch.no_grad():
        for _ in range(10):  # lazy-init costs (tables, spectra, workspaces)
            model(torch.rand(4, 1, 250, device=device), torch.rand(4, 2, 250, device=device))
        for bs in BUFFER_SIZES:
            T = bs // 128
            f0 = torch.rand(batch_size, 1, T, device=device)
            control = torch.rand(batch_size, 2, T, device=device)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    model(f0, control)
            torch.cuda.current_stream().wait_stream(side)
            graph = None
            if not no_graph:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = model(f0, control)
            run = graph.replay if graph is not None else (lambda: model(f0, control))
            for _ in range(20):
                run()
            torch.cuda.synchronize()
            lat = []
            for _ in range(num_iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                e1.synchronize()
                lat.append(e0.elapsed_time(e1) * 1e-3)
                rows.append([model_name, "gpu", bs, lat[-1]])
            lat = np.array(lat)
            summary[bs] = (np.percentile(lat, 50), np.percentile(lat, 99))
            print(f"buffer {bs:6d} samples ({bs / 16.0:8.2f} ms of audio): p50 {summary[bs][0] * 1e6:8.1f} us  "
                  f"p99 {summary[bs][1] * 1e6:8.1f} us  -> {bs / 16000.0 / summary[bs][0]:8.1f}x real-time "
                  f"({'hipGraph' if graph is not None else 'eager'})")
    if output_file:
        with open(output_file, "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["", "0", "1", "2", "3"])
            for i, r in enumerate(rows):
                wr.writerow([i] + r)
    return summary


if __name__ == "__main__":
    main()
