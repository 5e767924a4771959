#!/usr/bin/env python
"""Ablation timing of exciter_newt_kernel (B=64, T=500): which phase dominates?  GPU only."""
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
B, T = int(os.environ.get("B", 64)), 500
torch.manual_seed(0)
kind = os.environ.get("F0", "rand")
f0 = torch.rand(B, T, device="cuda") if kind == "rand" else 100 + 900 * torch.rand(B, 1, device="cuda").expand(B, T).contiguous()
control = torch.rand(B, 2, T, device="cuda")
eng = m._engine
w, _, _ = eng.weights()
carry = eng.phase_carry(f0=f0)
gru = eng.control_gru(control)
_, film, _, _ = eng.frame_mlps(gru)
pu = torch.rand(101, device="cuda")
out = torch.empty(B, 128 * T, device="cuda")
names = {0: "product", 1: "no sin", 2: "no LUT gather", 3: "no shaper tail", 4: "no MFMA"}
for v in (0, 1, 2, 3, 4, 0):
    def run():
        _lib.check(_lib.lib().nws_debug_exciter_newt(v, C.byref(w), f0.data_ptr(), carry.data_ptr(), pu.data_ptr(),
                                                     eng.rand_phase().data_ptr(), film.data_ptr(), B, T, 16000.0,
                                                     out.data_ptr(), _lib.stream_ptr()))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    e1.synchronize()
    print(f"variant {v} ({names[v]:>15s}): {e0.elapsed_time(e1) / 10:.4f} ms   [F0={kind}, B={B}]")
