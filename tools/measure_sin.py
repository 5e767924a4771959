#!/usr/bin/env python
"""Accuracy and throughput of the candidate sine implementations on the GPU (diagnostics)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_lib = importlib.import_module("neural-waveshaping-synthesis_amd._lib")
rng = np.random.default_rng(0)
names = {0: "nws_sinf (shipped)", 1: "v_sin_f32 + exact turns", 2: "odd poly + exact turns"}
for scale in (1e1, 1e3, 1e5, 5e6):
    x = rng.uniform(-scale, scale, 1 << 20).astype(np.float32)
    xd = torch.from_numpy(x).cuda(); y = torch.empty_like(xd)
    ref = np.sin(x.astype(np.float64))
    for mode in (0, 1, 2):
        _lib.check(_lib.lib().nws_debug_sin(mode, xd.data_ptr(), y.data_ptr(), xd.numel(), 1, _lib.stream_ptr()))
        e = np.abs(y.cpu().numpy().astype(np.float64) - ref)
        print(f"scale {scale:8.0e} mode {mode} {names[mode]:26s} max_abs_err {e.max():.3e} rms {np.sqrt((e**2).mean()):.3e}")
x = torch.from_numpy(rng.uniform(-1e4, 1e4, 1 << 22).astype(np.float32)).cuda(); y = torch.empty_like(x)
for mode in (0, 1, 2, 0):
    for _ in range(2):
        _lib.check(_lib.lib().nws_debug_sin(mode, x.data_ptr(), y.data_ptr(), x.numel(), 64, _lib.stream_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _lib.check(_lib.lib().nws_debug_sin(mode, x.data_ptr(), y.data_ptr(), x.numel(), 64, _lib.stream_ptr()))
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"throughput mode {mode} {names[mode]:26s}: {ms:.3f} ms for {x.numel()*64/1e6:.0f} M sines -> {x.numel()*64/ms/1e6:.1f} G sin/s")
