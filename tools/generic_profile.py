"""The runtime-size path (csrc/generic.hip) timed at the DEFAULT sizes (a tilted FIR window makes the default model leave the fused
kernels) next to the fused path: `python tools/generic_profile.py [B T]...`; under `rocprofv3 --kernel-trace --stats` it gives the
per-kernel breakdown quoted in DESIGN.md."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nws_amd as nws  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    nws.ensure_default_config()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck = os.path.join(root, "tests", "golden", "weights_vn.npz")
    shapes = [(1, 500), (64, 500)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    for fast in (True, False):
        fused = nws.NeuralWaveshaping.load_from_checkpoint(ck).cuda().eval()
        gen = nws.NeuralWaveshaping.load_from_checkpoint(ck)
        sd = gen.state_dict()
        sd["noise_synth.window"] = torch.as_tensor((np.hanning(257)[:256] * np.linspace(0.2, 1.0, 256)).astype(np.float32))
        gen.load_state_dict(sd)
        gen = gen.cuda().eval()
        if fast:
            fused.newt = nws.FastNEWT(fused.newt)
            gen.newt = nws.FastNEWT(gen.newt)
        assert fused._engine.specialised() and not gen._engine.specialised()
        for B, T in shapes:
            f0 = 100.0 + 400.0 * torch.rand(B, 1, T, device="cuda")
            c = torch.randn(B, 2, T, device="cuda")
            with torch.no_grad():
                tf = timed(lambda: fused(f0, c), 10)
                tg = timed(lambda: gen(f0, c), 5 if B > 1 else 10)
            print(f"{'FastNEWT' if fast else 'exact   '} B {B:3d} T {T}: fused {tf:7.3f} ms   runtime-size {tg:7.3f} ms   ratio {tg / tf:5.1f}", flush=True)


def small_config():
    """the g8_small configuration (tests/golden/g8_small.npz: 60 harmonics, 32 shapers of width 16 / depth 3, GRU 96, embedding 80,
    hop 64, 128-tap FIR, two NEWT output channels, 1 s reverb) at B = 16, T = 500 - VERDICT r3 #7's measurement"""
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g8_small.npz"))
    nws.gin.clear_config()
    nws.gin.parse_config(str(z["__gin__"]))
    try:
        for fast in (True, False):
            m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                        "tests", "golden", "g8_small.npz")).cuda().eval()
            if fast:
                m.newt = nws.FastNEWT(m.newt, table_size=int(z["__table_size__"]), table_min=float(z["__table_min__"]),
                                      table_max=float(z["__table_max__"]))
            for B, T in ((1, 500), (16, 500), (64, 500)):
                f0 = 100.0 + 400.0 * torch.rand(B, 1, T, device="cuda")
                c = torch.randn(B, z["__control__"].shape[1], T, device="cuda")
                with torch.no_grad():
                    tg = timed(lambda: m(f0, c), 10)
                sec = B * T * int(m.control_hop) / float(m.sample_rate)
                print(f"g8_small {'FastNEWT' if fast else 'exact   '} B {B:3d} T {T}: runtime-size {tg:7.3f} ms = {sec / (tg * 1e-3):9.0f} x real-time", flush=True)
    finally:
        nws.gin.clear_config()
        nws.gin.parse_config_file(nws.DEFAULT_GIN)


if __name__ == "__main__":
    main()
    if len(sys.argv) <= 2:
        small_config()
