#!/usr/bin/env python
"""Streaming hop, launch-structure A/B (same box, one process per variant): runs 40 hops of 2 frames with fixed inputs and an
injected noise stream, writes the emitted samples, and prints HIP-event p50 / p99 of 1000 graph-replayed hops.  The caller sets
NWS_STREAM_SPLIT_REVERB etc.; tools/stream_hop_ab.sh compares the dumps bit for bit."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, B = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dump_only = len(sys.argv) > 3 and sys.argv[3] == "dump-only"
    nws = importlib.import_module("neural-waveshaping-synthesis_amd")
    nws.ensure_default_config()
    model = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(ROOT, "tests", "golden", "weights_vn.npz")).cuda().eval()
    model.newt = nws.FastNEWT(model.newt)
    g = torch.Generator(device="cpu").manual_seed(7)
    # NWS_AB_CHUNKS="2,1,7,2,2,3" replaces the 40 hops of two frames by that chunk sequence (differential runs of the launch structures)
    chunks = [int(c) for c in os.environ["NWS_AB_CHUNKS"].split(",")] if os.environ.get("NWS_AB_CHUNKS") else [2] * 40
    K, total = 2, sum(chunks)
    f0 = (220 + 20 * torch.rand(B, 1, total, generator=g)).cuda()
    control = torch.randn(B, 2, total, generator=g).cuda()
    noise = torch.rand(128 * total - 1, generator=g).cuda()
    phase_u = torch.rand(1, 101, 1, generator=g).cuda()
    outs = []
    with torch.no_grad():
        s = model.stream(B, phase_u=phase_u, noise=noise, graph=True)
        at = 0
        for i, k in enumerate(chunks):
            outs.append(s.push(f0[:, :, at:at + k], control[:, :, at:at + k], final=(i == len(chunks) - 1)).cpu().numpy())
            at += k
        np.save(out_path, np.concatenate(outs, axis=1))
        if dump_only:
            return
        s = model.stream(B, graph=True)
        f0h, ch = f0[:, :, :K].contiguous(), control[:, :, :K].contiguous()
        for _ in range(30):
            s.push(f0h, ch)
        f0_in, c_in, _ = s.static_io(K)
        f0_in.copy_(f0h[:, 0])
        c_in.copy_(ch)
        torch.cuda.synchronize()
        import time
        modes = ("hop", "push") + (("eager",) if os.environ.get("NWS_AB_EAGER") else ())
        g, f0_g, c_g, nz_g, out_g, pre_g = s._graphs[(K, 2)]
        for mode in modes:
            lat, host = [], []
            for _ in range(1000):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                t0 = time.perf_counter()
                if mode == "hop":
                    s.hop(K)
                elif mode == "push":
                    s.push(f0h, ch)
                else:          # the captured hop's own buffers, launched kernel by kernel: what the graph saves or costs on the host
                    nz_g.uniform_()
                    s._step(f0_g, c_g, False, False, nz_g, out_g, pre_g)
                    s._advance(K, 128 * K, False, False)
                host.append((time.perf_counter() - t0) * 1e6)
                e1.record()
                e1.synchronize()
                lat.append(e0.elapsed_time(e1) * 1e3)
            lat = np.array(lat)
            print(f"{os.environ.get('NWS_AB_LABEL', '?'):>14s} B={B} {mode:5s}: p50 {np.percentile(lat, 50):6.1f} us  p99 {np.percentile(lat, 99):6.1f} us"
                  f"   (host call p50 {np.percentile(host, 50):5.1f} us)")


if __name__ == "__main__":
    main()
