"""Frame-MLP kernel families timed back to back on one box: mode 1 = tile kernels (frame_mlps16 / 64), mode 2 = wave-resident
frames (round 4).  `python tools/mlp_variants.py [B T]...`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nws_amd as nws  # noqa: E402
from nws_amd import _lib  # noqa: E402


def main():
    nws.ensure_default_config()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(root, "tests", "golden", "weights_vn.npz")).cuda().eval()
    shapes = [(64, 500), (32, 500), (16, 500), (128, 500), (8, 4000)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    L = _lib.lib()
    MODES = tuple(int(x) for x in os.environ.get("MODES", "1,2").split(","))
    # only the product kernel families are compared (1 tile kernels, 2 wave-resident frames): an ablation mode's output is
    # meaningless by construction and used to print `nan` into the one correctness column of profiles/r0x/mlp_variants.txt
    if not set(MODES) <= {1, 2}:
        raise SystemExit("MODES: 1 (tile kernels) and / or 2 (wave-resident frames); timing ablations live in tools/mlp_timeline.py")
    for B, T in shapes:
        gru = torch.tanh(torch.randn(B, T, 128, device="cuda"))
        res = {}
        for mode in MODES + MODES:
            L.nws_debug_frame_mlps_kernel(mode)
            for _ in range(5):
                out = m._engine.frame_mlps(gru)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                out = m._engine.frame_mlps(gru)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) * 10.0)
            res[("out", mode)] = out
        L.nws_debug_frame_mlps_kernel(0)
        a, b = MODES[0], MODES[-1]
        d_film = float((res[("out", a)][1] - res[("out", b)][1]).abs().max())
        d_fir = float((res[("out", a)][3] - res[("out", b)][3]).abs().max())
        print(f"B {B:4d} T {T:5d} frames {B * T:7d}: " + "  ".join(f"mode {k}: {min(res[k]):7.1f} us" for k in MODES) +
              f"   first vs last: max |film diff| {d_film:.2e} |fir diff| {d_fir:.2e}", flush=True)


if __name__ == "__main__":
    main()
