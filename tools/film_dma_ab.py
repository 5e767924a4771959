#!/usr/bin/env python
"""Oscillator kernel with the FiLM rows as per-frame fragment records by LDS-DMA (kOptFilmDma) against the
staging arithmetic it replaces: prologue-only launches (DBG 5: variants 5 / 6) and the whole kernel (variants 44 / 108), one
stream, same box; outputs compared.  GPU only.  -> profiles/r06/film_dma_ab.txt"""
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
VARIANTS = tuple(int(v) for v in os.environ.get("VARIANTS", "5,6,44,108").split(","))
eng = m._engine
w, _, _ = eng.weights()
L = _lib.lib()
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
    frags = torch.empty(B * T * (1536 + 16), dtype=torch.uint8, device="cuda")
    _lib.check(L.nws_debug_film_frags(C.byref(w), film.data_ptr(), B, T, frags.data_ptr(), _lib.stream_ptr()))
    out = {v: torch.zeros(B, 128 * T, device="cuda") for v in VARIANTS}

    def run(v):
        src = frags if v in (6, 108) else film
        _lib.check(L.nws_debug_exciter_newt(v, C.byref(w), f0.data_ptr(), carry.data_ptr(), pu.data_ptr(), eng.rand_phase().data_ptr(),
                                            src.data_ptr(), B, T, 16000.0, out[v].data_ptr(), _lib.stream_ptr()))

    times = {v: [] for v in out}
    for rnd in range(5):
        for v in out:
            for _ in range(3):
                run(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(v)
            e1.record()
            e1.synchronize()
            times[v].append(e0.elapsed_time(e1) / 20 * 1e3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _lib.check(L.nws_debug_film_frags(C.byref(w), film.data_ptr(), B, T, frags.data_ptr(), _lib.stream_ptr()))
    e1.record()
    e1.synchronize()
    names = {5: "prologue only, staging arithmetic", 6: "prologue only, records by LDS-DMA", 44: "whole kernel, staging arithmetic (product)",
             108: "whole kernel, records by LDS-DMA"}
    for v in out:
        print(f"[{kind}] {names[v]:46s} min {min(times[v][1:]):7.1f} us   all {['%.1f' % t for t in times[v]]}")
    if 108 in out:
        print(f"[{kind}] conversion kernel (fp32 rows -> records, stand-in for the frame-MLP output stage) {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    for v in out:
        if v in (108, 1044) and 44 in out:
            d = out[v].double() - out[44].double()
            scale = float(out[44].double().pow(2).mean().sqrt())
            print(f"[{kind}] variant {v} vs product: max|d| {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e} (signal rms {scale:.3e}) "
                  f"bit-equal {bool(torch.equal(out[v], out[44]))}")
