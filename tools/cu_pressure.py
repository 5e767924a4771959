#!/usr/bin/env python
"""CU-pressure table (VERDICT r5 #1c): the plain pipelined step (B = 64 x 4 s) with a resident kernel of N workgroups x 256 threads
busy for ~0.3 ms of every step on the pipeline's exchange stream - what a collective's ring kernels (RCCL: one workgroup per
channel) take from the chip while the 115 MB of a step's rows arrive per rank at 8 GPUs.  One process, one pipeline, every N
timed on the same streams; N = 0 is the plain step, 'launch only' the same launch with a 1 us kernel.

    python tools/cu_pressure.py [--steps 200] [--busy-us 300] > profiles/r06/cu_pressure.txt
"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--busy-us", type=int, default=300)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=500)
    a = ap.parse_args()
    import bench
    import nws_amd as nws
    _lib = nws._lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    nws.ensure_default_config()
    model = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(ROOT, "tests", "golden", "weights_vn.npz")).to(dev).eval()
    model.newt = nws.FastNEWT(model.newt)
    a.inputs = "rand"
    f0, control = bench.make_inputs(a, dev, 0)
    pipe = nws.pipeline.ForwardPipeline(model, depth=4, audio_streams=2, control_streams=2)
    sink = torch.zeros(256, device=dev)
    xs = pipe.exchange

    def region(groups, busy_us, steps):
        def step():
            pipe.submit(f0, control)
            if groups:
                _lib.check(L.nws_debug_queue_busy(groups, busy_us, sink.data_ptr(), xs.cuda_stream), "busy")
        for _ in range(80):
            step()
        pipe.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        pipe.synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    rep = nws.pipeline.placement_report(dev)
    print(f"# CU-pressure table: plain pipelined step, B = {a.batch} x {a.frames} frames, {a.steps} steps per line; a kernel of N x 256 threads busy for "
          f"{a.busy_us} us launched once per step on the exchange stream (tools/cu_pressure.py)")
    print(f"# placement: ok={rep['ok']} verified={rep.get('verified')} queue_offset={rep.get('queue_offset')}")
    with torch.no_grad():
        base = region(0, 0, a.steps)
        print(f"{'N = 0 (plain step)':28s} {base:.4f} ms/step  x1.000")
        t = region(1, 1, a.steps)
        print(f"{'launch only (1 x 1 us)':28s} {t:.4f} ms/step  x{t / base:.3f}")
        for n in (4, 8, 16, 32, 64):
            t = region(n, a.busy_us, a.steps)
            print(f"{'N = %d x %d us' % (n, a.busy_us):28s} {t:.4f} ms/step  x{t / base:.3f}   ({n * 4} waves on {n} CUs, busy {a.busy_us / (t * 1e3) * 100:.0f} % of the step)")
        t = region(0, 0, a.steps)
        print(f"{'N = 0 again':28s} {t:.4f} ms/step  x{t / base:.3f}")


if __name__ == "__main__":
    main()
