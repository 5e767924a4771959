#!/usr/bin/env python
"""Print which command-processor pipe every hardware queue of this process sits on, as seen by nws_queue_probe (the
measurement behind pipeline.placed_streams): n streams first used one after the other, then the matrix
frac[hold stream][touch stream] (>= ~1: the touch waited for the hold grid to be handed out = same pipe; ~0: served meanwhile).

    python tools/queue_pipe_map.py [--streams 10] [--groups 8192] [--spin-us 10] [--high 0]
"""
import argparse
import ctypes as C
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=10)
    ap.add_argument("--groups", type=int, default=8192)
    ap.add_argument("--spin-us", type=int, default=10)
    ap.add_argument("--high", type=int, default=0, help="every k-th stream high priority (0: none)")
    ap.add_argument("--layout", default="", help="creation order as a string of n / h (normal / high priority), overrides --streams / --high")
    ap.add_argument("--repeat", type=int, default=1)
    a = ap.parse_args()
    import nws_amd
    _lib = nws_amd._lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    scratch = torch.zeros(4, dtype=torch.int64, device=dev)
    touch = torch.zeros(64, device=dev)
    streams = [("null", torch.cuda.current_stream(dev))]
    for k in range(len(a.layout) or a.streams):
        hi = (a.layout[k] == "h") if a.layout else (a.high and (k % a.high == a.high - 1))
        s = torch.cuda.Stream(device=dev, priority=-1 if hi else 0)
        with torch.cuda.stream(s):
            touch.fill_(0.0)
        s.synchronize()
        streams.append((f"s{k}{'h' if hi else ''}", s))
    frac = C.c_float()
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} groups={a.groups} spin_us={a.spin_us}")
    for rep in range(a.repeat):
      print("hold\\touch " + " ".join(f"{n:>6s}" for n, _ in streams))
      for hn, hs in streams:
        row = []
        for tn, ts in streams:
            if hs is ts:
                row.append("     -")
                continue
            _lib.check(L.nws_queue_probe(hs.cuda_stream, ts.cuda_stream, a.groups, a.spin_us, scratch.data_ptr(), C.byref(frac)))
            row.append(f"{frac.value:6.2f}" if frac.value > 0.3 else "     .")
        print(f"{hn:>10s} " + " ".join(row))


if __name__ == "__main__":
    main()
