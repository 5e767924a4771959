#!/usr/bin/env python
"""Timing + agreement of the fused oscillator/NEWT kernel's compile-time options (nws_debug_exciter_newt variants 10 + OPT
bits: 1 scalar sines, 2 FiLM interpolation on the matrix pipe, 4 one-term fp16 sines, 8 + 16 hybrid products, 32 the 80-register
tail; 12 = the round-3 default, 44 = the default, 36 / 68 = the hybrid-W opt-in without / with the 80-register tail) at B=64, T=500.  GPU only."""
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
m.newt = nws.FastNEWT(m.newt)
B, T = int(os.environ.get("B", 64)), int(os.environ.get("T", 500))
variants = [int(v) for v in os.environ.get("VARIANTS", "12,44,36,68").split(",")]
eng = m._engine
w, _, _ = eng.weights()
for kind in ("rand", "real"):
    torch.manual_seed(0)
    if kind == "rand":
        f0 = torch.rand(B, T, device="cuda")
        control = torch.rand(B, 2, T, device="cuda")
    else:
        tt = torch.arange(T, device="cuda") * (128.0 / 16000.0)
        f0 = ((100 + 900 * torch.rand(B, 1, device="cuda")) * (1 + 0.01 * torch.sin(2 * torch.pi * 5.5 * tt))).contiguous()
        control = torch.randn(B, 2, T, device="cuda")
    carry = eng.phase_carry(f0=f0)
    gru = eng.control_gru(control)
    _, film, _, _ = eng.frame_mlps(gru)
    pu = torch.rand(101, device="cuda")
    _, ref = eng.exciter_newt(f0, None, carry, pu, film)
    torch.cuda.synchronize()
    scale = float(ref.double().pow(2).mean().sqrt())
    out = torch.empty(B, 128 * T, device="cuda")

    def run(v):
        _lib.check(_lib.lib().nws_debug_exciter_newt(v, C.byref(w), f0.data_ptr(), carry.data_ptr(), pu.data_ptr(),
                                                     eng.rand_phase().data_ptr(), film.data_ptr(), B, T, 16000.0,
                                                     out.data_ptr(), _lib.stream_ptr()))

    times = {v: [] for v in variants}
    errs = {}
    for rnd in range(4):
        for v in variants:
            for _ in range(3):
                run(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(v)
            e1.record()
            e1.synchronize()
            times[v].append(e0.elapsed_time(e1) / 20)
            if rnd == 0:
                d = (out.double() - ref.double())
                errs[v] = (float(d.abs().max()), float(d.pow(2).mean().sqrt()))
    for v in variants:
        print(f"[{kind}] variant {v} (OPT={v - 10 if v >= 10 else '-'}): min {min(times[v][1:]):.4f} ms  all {['%.4f' % t for t in times[v]]}  "
              f"vs product: max|d| {errs[v][0]:.3e} rms {errs[v][1]:.3e} (signal rms {scale:.3e})")
