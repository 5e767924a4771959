#!/usr/bin/env python
"""Where the time of a SHORT timed region goes: completion time of every batch of a K-step pipelined run (GPU only)."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
nws = importlib.import_module("neural-waveshaping-synthesis_amd")
nws.ensure_default_config()
m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(ROOT, "tests/golden/weights_vn.npz")).cuda().eval()
m.newt = nws.FastNEWT(m.newt)
K = int(os.environ.get("K", 20))
cs = int(os.environ.get("CS", 2))
f0, control = torch.rand(64, 1, 500, device="cuda"), torch.rand(64, 2, 500, device="cuda")
pipe = nws.ForwardPipeline(m, depth=int(os.environ.get("DEPTH", 4)), audio_streams=int(os.environ.get("AS", 2)), control_streams=cs,
                           chain_exciters=os.environ.get("CHAIN", "0") == "1")
with torch.no_grad():
    for _ in range(125):
        pipe.submit(f0, control)
    pipe.synchronize()
    torch.cuda.synchronize()
    for rep in range(3):
        start = torch.cuda.Event(enable_timing=True)
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        start.record()
        host = []
        for i in range(K):
            au = pipe.next_audio_stream()
            pipe.submit(f0, control)
            ends[i].record(au)
            host.append((time.perf_counter() - t0) * 1e3)
        pipe.synchronize()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        done = [start.elapsed_time(e) for e in ends]
        print(f"rep {rep}: wall {wall:.3f} ms = {wall / K:.4f} ms/step; host submit done at {host[-1]:.3f} ms")
        print("   batch completion (ms):", " ".join(f"{d:.2f}" for d in done))
        print("   host submit times (ms):", " ".join(f"{h:.2f}" for h in host))
