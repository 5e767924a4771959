"""Cycle timeline of the two workgroups (newt.mlp / h_generator path) of frame_mlps_few_kernel on two frames of one utterance
(s_memtime probes of wave 0, csrc/mlp_few.h): entry | fragments requested | input staged | proj | hidden 1-3 | output(s) [| FIR]."""
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
B, T = 1, 2
gru = torch.tanh(torch.randn(B, T, 128, device="cuda"))
buf = torch.zeros(32, dtype=torch.int64, device="cuda")
for rep in range(6):
    buf.zero_()
    torch.cuda.synchronize()
    L.nws_debug_frame_mlps_probe(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    m._engine.frame_mlps(gru)
    e1.record()
    torch.cuda.synchronize()
    L.nws_debug_frame_mlps_probe(0)
    t = buf.cpu().reshape(2, 16)
    for path in range(2):
        row = t[path]
        n = int((row != 0).sum())
        d = (row[1:n] - row[:n - 1]).tolist()
        print(f"rep {rep} path {path}: total {int(row[n - 1] - row[0])} ticks cycles, deltas {d}; events {e0.elapsed_time(e1) * 1e3:.1f} us")
