#!/usr/bin/env python
"""Long bit-exact soak of ForwardPipeline (default: two audio streams) on the bench's shape: every batch of every round is
compared with the plain forward of the same inputs and draws.  Prints one JSON line.

    python tools/soak_pipeline.py [--rounds 200] [--distinct 12] [--audio-streams 2]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=200)
    ap.add_argument("--distinct", type=int, default=12)
    ap.add_argument("--audio-streams", type=int, default=2)
    ap.add_argument("--control-streams", type=int, default=1)
    a = ap.parse_args()
    import torch
    import nws_amd
    from gpu_util import build_model
    m = build_model(True)
    eng = m._engine
    B, T = 64, 500
    g = torch.Generator(device="cuda").manual_seed(11)
    batches, refs = [], []
    for _ in range(a.distinct):
        f0 = (100 + 900 * torch.rand(B, 1, 1, device="cuda", generator=g)) * (1 + 0.01 * torch.randn(B, 1, T, device="cuda", generator=g))
        c = torch.randn(B, 2, T, device="cuda", generator=g)
        pu = torch.rand(101, device="cuda", generator=g)
        nz = torch.rand(128 * T - 1, device="cuda", generator=g)
        batches.append((f0, c, pu, nz))
        ws = eng.new_workspace(B, T)
        eng.forward_control(f0, c, ws, batched_gru=False)
        refs.append(eng.forward_audio(f0, B, T, pu, nz, ws).clone())
    torch.cuda.synchronize()
    pipe = nws_amd.ForwardPipeline(m, audio_streams=a.audio_streams, control_streams=a.control_streams)
    bad = 0
    t0 = time.perf_counter()
    for r in range(a.rounds):
        outs = [pipe.submit(f0, c, phase_u=pu, noise=nz) for f0, c, pu, nz in batches]
        if r % 3 == 1:   # perturb the relative timing of the streams now and then
            with torch.cuda.stream(pipe.audio[r % len(pipe.audio)]):
                torch.empty(1 << 20, device="cuda").normal_()
        pipe.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o, ref in zip(outs, refs))
    print(json.dumps({"soak": "ForwardPipeline vs plain forward, bit for bit", "audio_streams": a.audio_streams,
                      "control_streams": a.control_streams, "batches": a.rounds * a.distinct, "mismatching": bad,
                      "seconds": round(time.perf_counter() - t0, 2)}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
