"""Per-length cost of the reverb stage and of the whole forward: direct plans against overlap-save plans of every block size
(NWS_REVERB_OLS_N2 pins the row size).  Prints one line per (T, B, block size); INTEGRATION.md quotes the table."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nws_amd as nws  # noqa: E402
from nws_amd import _lib, engine as nws_engine  # noqa: E402


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
    m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(root, "tests", "golden", "weights_vn.npz")).cuda().eval()
    m.newt = nws.FastNEWT(m.newt)
    cases = [(500, 64), (501, 64), (504, 64), (1000, 64), (1001, 64), (2000, 16), (2001, 16), (8193, 8), (9375, 8), (37500, 2), (37501, 2)]
    for T, B in cases:
        N = 128 * T
        x = torch.randn(B, N, device="cuda")
        f0 = 200.0 + 300.0 * torch.rand(B, 1, T, device="cuda")
        ctl = torch.randn(B, 2, T, device="cuda")
        for n2 in (0, 512, 1024, 2048):
            if n2:
                os.environ["NWS_REVERB_OLS_N2"] = str(n2)
            else:
                os.environ.pop("NWS_REVERB_OLS_N2", None)
            nws_engine._PLAN_CACHE.clear()
            m._engine._workspaces.clear()
            plan = _lib.NwsReverbPlan()
            _lib.lib().nws_reverb_plan(N, 32000, C.byref(plan))
            if n2 and plan.Lc == 0:
                continue
            it = 20 if N * B < 3e7 else 5
            t_rev = timed(lambda: m._engine.reverb(x), it)
            with torch.no_grad():
                t_fwd = timed(lambda: m(f0, ctl), max(2, it // 4))
            print(f"T {T:6d} ({N / 16000.0:7.1f} s) B {B:3d} plan L {plan.L:7d} = {plan.N1} x {plan.N2} "
                  f"{'direct' if plan.Lc == 0 else 'ols x%d' % plan.nblk:9s} {'(auto)' if not n2 else '(pinned)':8s} "
                  f"reverb {t_rev * 1e3:9.1f} us = {t_rev * 1e6 / (B * N / 16000.0):7.3f} us per second of audio; forward {t_fwd:8.3f} ms", flush=True)
    os.environ.pop("NWS_REVERB_OLS_N2", None)


if __name__ == "__main__":
    main()
