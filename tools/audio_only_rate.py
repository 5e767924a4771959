#!/usr/bin/env python
"""What the control half costs the pipelined step: the audio halves alone (frame MLPs, oscillator + NEWT, noise, reverb) of
prepared batches alternating over two audio streams, with pieces of the control half running beside them on a side stream
WITHOUT dependencies.  One configuration per process (HIP maps streams onto a few hardware queues; stale streams of an
earlier configuration would share them):  WHAT=none|rng|gru|grub|grubT|all python tools/audio_only_rate.py.  GPU only.
Measured (MI355X, B=64, T=500): audio halves alone 0.3455 ms/step on two streams (0.375 on one, 0.356 on three); with the
per-utterance recurrence + carries beside them 0.3833-0.386; with the two RNG draws as well 0.387 - i.e. the control half
costs the pipelined step ~40 us although it is off the critical path: its 64 workgroups hold 248 of the 512 registers of
every SIMD of 64 CUs for 0.23 of the 0.38 ms, where only two waves of the oscillator kernel fit beside them."""
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
B, T, K = 64, 500, int(os.environ.get("K", 200))
what = os.environ.get("WHAT", "none")
NSLOT = int(os.environ.get("NSLOT", 4))
n_audio = int(os.environ.get("AUDIO_STREAMS", 2))
eng = m._engine
import ctypes as C
_lib = importlib.import_module("neural-waveshaping-synthesis_amd._lib")
wdesc, _, _ = eng.weights()
gru_t = int(os.environ.get("GRU_T", 250))
torch.manual_seed(0)
f0 = torch.rand(B, 1, T, device="cuda")
control = torch.rand(B, 2, T, device="cuda")
pu = torch.rand(101, device="cuda")
nz = torch.rand(128 * T - 1, device="cuda")
cshort = torch.rand(B, 2, gru_t, device="cuda")
gshort = [torch.empty(B, gru_t, 128, device="cuda") for _ in range(4)]
with torch.no_grad():
    streams = [torch.cuda.Stream() for _ in range(n_audio)]
    n_side = int(os.environ.get("SIDE_STREAMS", 1))
    sides = [torch.cuda.Stream(priority=int(os.environ.get("PRIO", -1))) for _ in range(n_side)] if what != "none" else None
    side = sides[0] if sides else None
    slots = [eng.new_workspace(B, T) for _ in range(4)]
    spare = [eng.new_workspace(B, T) for _ in range(4)]
    for ws in slots + spare:
        eng.forward_control(f0, control, ws)
    outs = [torch.empty(B, 128 * T, device="cuda") for _ in slots]
    torch.cuda.synchronize()
    delay_us = float(os.environ.get("DELAY_US", 0))   # once per repetition: stream 1 starts this much later (phase probe)
    for rep in range(4):
        if delay_us and n_audio > 1:
            with torch.cuda.stream(streams[1]):
                torch.cuda._sleep(int(delay_us * 2400))
        t0 = time.perf_counter()
        for i in range(K):
            if side is not None:
                with torch.cuda.stream(sides[i % n_side]):
                    if what in ("rng", "all"):
                        torch.rand(101, device="cuda")
                        torch.rand(128 * T - 1, device="cuda")
                    if what in ("gru", "all"):
                        eng.forward_control(f0, control, spare[i % 4], batched_gru=False)
                    if what == "grub":      # the batched MFMA recurrence instead: 4 workgroups for 64 utterances
                        eng.forward_control(f0, control, spare[i % 4], batched_gru=True)
                    if what == "grubT":     # the same on GRU_T frames only: what a recurrence of that much less CU time would cost
                        _lib.check(_lib.lib().nws_control_gru_batched(C.byref(wdesc), cshort.data_ptr(), B, 2, gru_t, None,
                                                                      gshort[i % 4].data_ptr(), None, _lib.stream_ptr()))
            with torch.cuda.stream(streams[i % n_audio]):
                eng.forward_audio(f0, B, T, pu, nz, slots[i % NSLOT], out=outs[i % NSLOT])
        host = (time.perf_counter() - t0) / K * 1e3
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / K * 1e3
    print(f"{n_audio} audio stream(s), {n_side if side is not None else 0} side stream(s): {what}: {el:.4f} ms/step (host submit {host:.4f} ms/step)")
