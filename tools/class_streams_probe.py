#!/usr/bin/env python
"""Audio half issued by KERNEL CLASS instead of by batch: stream X runs the oscillator kernels of consecutive batches back to
back, stream Y the frame MLPs, stream Z noise + reverb, chained per batch by events - so the VALU-bound oscillator kernel
always has the latency-bound kernels of neighbouring batches beside it.  Compared with whole audio halves alternating over
two streams (what ForwardPipeline does).  Control halves prepared beforehand (no GRU here).  GPU only."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nws = importlib.import_module("neural-waveshaping-synthesis_amd")
nws.ensure_default_config()
m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(ROOT, "tests/golden/weights_vn.npz")).cuda().eval()
m.newt = nws.FastNEWT(m.newt)
eng = m._engine
B, T, K = 64, 500, int(os.environ.get("K", 200))
mode = os.environ.get("MODE", "class")
torch.manual_seed(0)
f0 = torch.rand(B, T, device="cuda")
control = torch.rand(B, 2, T, device="cuda")
pu = torch.rand(101, device="cuda")
nz = torch.rand(128 * T - 1, device="cuda")
with torch.no_grad():
    carry = eng.phase_carry(f0=f0)
    gru = eng.control_gru(control)
    torch.cuda.synchronize()
    if mode == "class":
        sx, sy, sz = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    else:
        ss = [torch.cuda.Stream(), torch.cuda.Stream()]

    def issue(i):
        if mode == "class":
            with torch.cuda.stream(sy):
                _, film, _, fir = eng.frame_mlps(gru)
                e1 = sy.record_event()
            with torch.cuda.stream(sx):
                sx.wait_event(e1)
                _, newt = eng.exciter_newt(f0, None, carry, pu, film)
                film.record_stream(sx)
                e2 = sx.record_event()
            with torch.cuda.stream(sz):
                sz.wait_event(e2)
                pre = eng.fir_noise(fir, nz, newt)
                fir.record_stream(sz)
                newt.record_stream(sz)
                return eng.reverb(pre)
        with torch.cuda.stream(ss[i % 2]):
            _, film, _, fir = eng.frame_mlps(gru)
            _, newt = eng.exciter_newt(f0, None, carry, pu, film)
            pre = eng.fir_noise(fir, nz, newt)
            return eng.reverb(pre)

    for i in range(20):
        issue(i)
    torch.cuda.synchronize()
    for rep in range(4):
        t0 = time.perf_counter()
        for i in range(K):
            y = issue(i)
        host = (time.perf_counter() - t0) / K * 1e3
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / K * 1e3
    print(f"mode {mode}: {el:.4f} ms/step (host submit {host:.4f})")
