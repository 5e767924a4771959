#!/usr/bin/env python
"""Timing of control_gru_kernel and its ablation variants (nws_debug_control_gru; see include/nws_hip.h) at B=64 and B=1,
T=500.  Variant 0 is the product kernel; 1-5 return wrong values by design; 6 is the product arithmetic with s_memtime
probes: its cycle timeline of steps 200..207 is printed last.  GPU only."""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nws = importlib.import_module("neural-waveshaping-synthesis_amd")
_lib = importlib.import_module("neural-waveshaping-synthesis_amd._lib")
nws.ensure_default_config()
m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(ROOT, "tests/golden/weights_vn.npz")).cuda().eval()
T = int(os.environ.get("T", 500))
variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3,4,5").split(",")]
w, _, _ = m._engine.weights()
for B in (64, 1):
    torch.manual_seed(0)
    control = torch.randn(B, 2, T, device="cuda")
    out = torch.empty(B, T, 128, device="cuda")
    ref = m._engine.control_gru(control)
    torch.cuda.synchronize()

    def run(v):
        _lib.check(_lib.lib().nws_debug_control_gru(v, C.byref(w), control.data_ptr(), B, 2, T, out.data_ptr(), _lib.stream_ptr()))

    for v in variants:
        ts = []
        for rnd in range(4):
            for _ in range(3):
                run(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(v)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        err = float((out - ref).abs().max())
        print(f"B={B} variant {v}: min {min(ts):.4f} ms = {min(ts) * 1e3 / T:.3f} us/step  max|d| vs product {err:.2e}")

    if T >= 208:
        _lib.check(_lib.lib().nws_debug_control_gru(6, C.byref(w), control.data_ptr(), B, 2, T, out.data_ptr(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        tk = out.view(-1)[:96].view(torch.int64).cpu().view(8, 6)
        d = (tk[:, 1:] - tk[:, :-1]).double()
        probe = float(d[:, 0].median())
        names = ("reads+FMAs", "reduction", "gates", "h store+barrier")
        print(f"B={B} timeline (cycles per phase, median of 8 steps, the probe's own {probe:.0f} removed): "
              + ", ".join(f"{n} {float(d[:, i + 1].median()) - probe:.0f}" for i, n in enumerate(names)))
