#!/usr/bin/env python
"""Timing ablations of the PRODUCT oscillator kernel (two hops per workgroup, FiLM on the matrix pipe, 80-register tail) at B=64, T=500:
nws_debug_exciter_newt variants 44 (product), 21 no sines, 22 no table gathers, 23 no tail, 24 no MFMAs, 5 prologue only, 26 the whole kernel without a global load in front of its barrier.  Results of the
ablations are wrong by design; what counts is the time each part takes out of the launch.  GPU only."""
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
eng = m._engine
w, _, _ = eng.weights()
names = {44: "product", 21: "no sines", 22: "no table gathers", 23: "no tail", 24: "no MFMAs", 5: "prologue only", 26: "no loads before the barrier", 27: "no FiLM-row loads", 28: "no fragment DMA", 29: "no F0 / carry / shift loads"}
if os.environ.get("ONLY"):
    names = {int(v): names[int(v)] for v in os.environ["ONLY"].split(",")}
for kind in os.environ.get("KINDS", "rand,real").split(","):
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
    out = torch.empty(B, 128 * T, device="cuda")

    def run(v):
        _lib.check(_lib.lib().nws_debug_exciter_newt(v, C.byref(w), f0.data_ptr(), carry.data_ptr(), pu.data_ptr(),
                                                     eng.rand_phase().data_ptr(), film.data_ptr(), B, T, 16000.0,
                                                     out.data_ptr(), _lib.stream_ptr()))

    res = {}
    for rnd in range(int(os.environ.get("ROUNDS", 3))):
        for v in names:
            for _ in range(3):
                run(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(v)
            e1.record()
            e1.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20)
    for v in names:
        print(f"[{kind}] variant {v:3d} ({names[v]:>16s}): min {min(res[v]):.4f} ms  all {['%.4f' % t for t in res[v]]}")
