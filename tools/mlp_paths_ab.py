"""Frame MLPs (frame_mlps_wr_kernel): do the two paths' workgroups cost each other anything?  (VERDICT r5 #5 asked for ONE workgroup
per frame block running both paths, proj computed once.)  Each path is a chain of six dependent phases (weight chunk -> MFMA burst ->
LayerNorm / split); newt.mlp and h_generator of a frame block run on two DIFFERENT CUs at the same time (250 workgroups on 256 CUs,
one 150 KB workgroup per CU).  Timed here, one stream, same box: both paths (the product), each path's workgroups alone (the other
path's workgroups leave at once), and both at half / double the batch.  A fused workgroup is the single-path launch's workgroup count
with eleven phases in a row instead of six: single-path time x 11 / 6 against the product's time says whether it could win.
-> profiles/r06/mlp_paths_ab.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nws_amd as nws  # noqa: E402
from nws_amd import _lib  # noqa: E402

nws.ensure_default_config()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(root, "tests", "golden", "weights_vn.npz")).cuda().eval()
L = _lib.lib()


def timed(mode, gru, reps=100):
    L.nws_debug_frame_mlps_kernel(mode)
    for _ in range(5):
        m._engine.frame_mlps(gru)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            m._engine.frame_mlps(gru)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    L.nws_debug_frame_mlps_kernel(0)
    return best


print("# frame_mlps_wr_kernel, one stream, 100 launches back to back (best of 4); tools/mlp_paths_ab.py")
for B, T in ((64, 500), (32, 500), (128, 500)):
    gru = torch.tanh(torch.randn(B, T, 128, device="cuda"))
    both = timed(2, gru)
    p0 = timed(2 + (7 << 8), gru)
    p1 = timed(2 + (8 << 8), gru)
    nblk = (B * T + 255) // 256
    print(f"B {B:3d} x T {T}: {nblk} frame blocks = {2 * nblk} workgroups   both paths {both:6.1f} us   newt.mlp workgroups alone {p0:6.1f} us   "
          f"h_generator workgroups alone {p1:6.1f} us   longer of the two {max(p0, p1):6.1f}   sum {p0 + p1:6.1f}")
