import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gpu_util import build_model
from oracle.newt_oracle import OracleNEWT, load_weights_npz
m = build_model(False); eng = m._engine
o = OracleNEWT(load_weights_npz('/root/repo/tests/golden/weights_vn.npz'), fast=False)
B, T = 17, 3
g = torch.Generator().manual_seed(100 * B + T)
H = (0.02 * torch.rand(B, 129, T, generator=g) ** 3 + 1e-4); H[0] *= 50; H[1] *= 1e-4
noise = torch.rand(128 * T - 1, generator=g)
ref = o.fir_noise(H, noise)[:, 0].numpy()
Ht = H.transpose(1, 2)
h = (torch.fft.irfft(torch.complex(Ht, torch.zeros_like(Ht))).roll(128, -1) * torch.hann_window(256).view(1, 1, -1)).contiguous().cuda()
out = eng.fir_noise(h, noise.cuda()).cpu().numpy()
small = torch.cat([eng.fir_noise(h[i:i + 8].contiguous(), noise.cuda()) for i in range(0, B, 8)]).cpu().numpy()
rowmax = np.abs(ref).max(1)
np.set_printoptions(precision=2, linewidth=200)
print('rowmax', rowmax)
print('mfma rel', np.abs(out - ref).max(1) / rowmax)
print('small rel', np.abs(small - ref).max(1) / rowmax)
e = np.abs(small - ref); r = 2
print('row', r, 'argmax', e[r].argmax(), e[r].max(), 'per-hop max', [e[r, 128*t:128*t+128].max() for t in range(T)])
