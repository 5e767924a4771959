"""Cycle timeline of one workgroup per path of frame_mlps_wr_kernel (s_memtime probes, ablation 6): where a wave's time goes."""
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
B, T = 64, 500
gru = torch.tanh(torch.randn(B, T, 128, device="cuda"))
buf = torch.zeros(2 * 8 * 32, dtype=torch.int64, device="cuda")
L.nws_debug_frame_mlps_probe(buf.data_ptr())
L.nws_debug_frame_mlps_kernel(2 + (6 << 8))
for _ in range(3):
    m._engine.frame_mlps(gru)
torch.cuda.synchronize()
L.nws_debug_frame_mlps_kernel(0)
t = buf.cpu().reshape(2, 8, 32)
print("probes: prologue | sync0 (entry, after barrier) | per layer: MFMA phase, epilogue, sync (entry, after barrier) ...")
for path in range(2):
    for wave in (0, 1, 4, 5):
        row = t[path, wave]
        n = int((row != 0).sum())
        d = (row[1:n] - row[:n - 1]).tolist()
        print(f"path {path} wave {wave}: t0 {int(row[0] - t[path, :, 0].min())} total {int(row[n - 1] - row[0])} deltas {d}")
