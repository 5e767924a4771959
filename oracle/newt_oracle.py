"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (PyTorch-CPU fp32, op for op) of the reference's
``NeuralWaveshaping.forward`` hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference from /root/reference in the build container and records golden
input/output vectors (``tests/golden/*.npz``); ``tests/test_oracle_golden.py``
checks this file against every one of them (bit-exact on the oscillator, LUT
and noise stages, <=1e-6 elsewhere).  The reference has no tests of its own
 so those recorded vectors are the pin.

Each function cites the reference lines it restates (paths relative to
/root/reference/neural_waveshaping_synthesis/).

It has no gin / pytorch-lightning dependency: weights come in as a plain
``{state_dict key: array}`` mapping, the two RNG draws the reference makes inside
forward can be injected (``phase_u``: the 101 U[0,1) draws of
models/modules/generators.py:55, ``noise``: the N-1 U[0,1) draws of
generators.py:30) or are drawn from torch's global CPU generator in the
reference's order.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

TAU = math.tau


def _t(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu()
    return torch.from_numpy(np.ascontiguousarray(x))


class OracleNEWT:
    def __init__(self, weights, sample_rate=16000, control_hop=128, fast=False,
                 table_size=4096, table_min=-3.0, table_max=3.0, lut_python_loop=True):
        self.w = {k: _t(v) for k, v in weights.items()}
        self.sample_rate = sample_rate
        self.control_hop = control_hop
        self.fast = fast
        self.table_size, self.table_min, self.table_max = table_size, table_min, table_max
        self.lut_python_loop = lut_python_loop
        w = self.w
        self.n_harmonics = w["harmonic_mixer.weight"].shape[1]
        self.n_waveshapers = w["harmonic_mixer.weight"].shape[0]
        hidden = w["embedding.gru.weight_hh_l0"].shape[1]
        # models/neural_waveshaping.py:21  nn.GRU(control_size, hidden_size, batch_first=True)
        self.gru = torch.nn.GRU(w["embedding.gru.weight_ih_l0"].shape[1], hidden, batch_first=True)
        with torch.no_grad():
            for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                getattr(self.gru, n).copy_(w["embedding.gru." + n])
        self.gru.eval()
        self.harmonic_axis = torch.arange(1, self.n_harmonics + 1).view(1, -1, 1)  # generators.py:47-48
        self.rand_phase = torch.ones(1, self.n_harmonics, 1) * TAU                 # generators.py:45
        self.window = w.get("noise_synth.window", torch.hann_window(256))          # generators.py:20
        self.ir_length = self.window.numel()
        self._table = None

    # ---- frame-rate path -------------------------------------------------
    def embedding(self, control):
        """models/neural_waveshaping.py:69-72 (get_embedding) + :24-26 (ControlModule.forward)."""
        f0, other = control[:, 0:1], control[:, 1:2]
        c = torch.cat((f0, other), dim=1)
        x, _ = self.gru(c.transpose(1, 2))
        self._gru_out = x
        return F.conv1d(x.transpose(1, 2), self.w["embedding.proj.weight"], self.w["embedding.proj.bias"])

    def td_mlp(self, x, prefix, depth=None):
        """models/modules/dynamic.py:20-40: [Conv1x1 -> LayerNorm(channels) -> LeakyReLU] x (depth-1) -> Conv1x1.
        depth: from the state dict's own keys (net.0, net.3, ...) unless given."""
        if depth is None:
            depth = sum(1 for i in range(64) if f"{prefix}.net.{3 * i}.weight" in self.w)
        for i in range(depth):
            k = f"{prefix}.net.{3 * i}"
            x = F.conv1d(x, self.w[k + ".weight"], self.w[k + ".bias"])
            if i < depth - 1:
                ln = f"{prefix}.net.{3 * i + 1}.layer_norm"
                # dynamic.py:16-17: layer norm over the channel axis via transpose
                x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), self.w[ln + ".weight"], self.w[ln + ".bias"]).transpose(1, 2)
                x = F.leaky_relu(x)
        return x

    # ---- exciter ----------------------------------------------------------
    def oscillator(self, f0_up, phase_u):
        """models/modules/generators.py:58-66 (HarmonicOscillator.forward).  f0_up: (B, N)."""
        phase = TAU * f0_up.cumsum(-1) / self.sample_rate                       # :59
        harmonic_phase = self.harmonic_axis * phase.unsqueeze(1)                # :60
        shift = phase_u.view(1, -1, 1) * self.rand_phase - math.pi              # :55
        harmonic_phase = harmonic_phase + shift                                 # :61
        mask = (f0_up.unsqueeze(1) * self.harmonic_axis) < (self.sample_rate / 2)  # :50-52
        self._phase = phase
        return torch.sin(harmonic_phase) * mask                                 # :64

    def exciter(self, f0_up, phase_u):
        """models/neural_waveshaping.py:64-67 (render_exciter)."""
        sig = self.oscillator(f0_up[:, 0], phase_u)
        self._osc = sig
        return F.conv1d(sig, self.w["harmonic_mixer.weight"], self.w["harmonic_mixer.bias"])

    # ---- shapers ----------------------------------------------------------
    def exact_shaper(self, x):
        """models/modules/shaping.py:36-37 (TrainableNonlinearity.forward), Sine activations; the depth is read off the state
        dict (net.0, net.2, ...: 4 layers for gin/models/newt.gin)."""
        w = self.w
        x = w["newt.shaping_fn.input_scale"] * x
        C = self.n_waveshapers
        for i in [2 * j for j in range(32) if f"newt.shaping_fn.net.{2 * j}.weight" in w]:
            x = torch.sin(F.conv1d(x, w[f"newt.shaping_fn.net.{i}.weight"], w[f"newt.shaping_fn.net.{i}.bias"], groups=C))
        return x

    def lookup_table(self):
        """models/modules/shaping.py:107-119 (FastNEWT._init_lookup_table)."""
        if self._table is None:
            sv = torch.linspace(self.table_min, self.table_max, self.table_size).expand(1, self.n_waveshapers, self.table_size)
            self._table = self.exact_shaper(sv)[0].contiguous()
        return self._table

    def _lookup(self, idx):
        table = self.lookup_table()
        if self.lut_python_loop:
            # models/modules/shaping.py:121-134: per (batch, shaper) index ops
            return torch.stack([torch.stack([table[s, idx[b, s]] for s in range(idx.shape[1])], 0)
                                for b in range(idx.shape[0])], 0)
        return torch.gather(table.unsqueeze(0).expand(idx.shape[0], -1, -1), 2, idx)

    def lut_shaper(self, x):
        """models/modules/shaping.py:136-151 (FastNEWT.shaping_fn), quirks kept."""
        idx = self.table_size * (x - self.table_min) / (self.table_max - self.table_min)
        lower = torch.floor(idx).long()
        lower[lower < 0] = 0
        lower[lower >= self.table_size] = self.table_size - 1
        upper = lower + 1
        upper[upper >= self.table_size] = self.table_size - 1
        fract = idx - lower
        lo = self._lookup(lower)
        up = self._lookup(upper)
        return (up - lo) * fract + lo

    def newt(self, exciter, emb):
        """models/modules/shaping.py:67-79 (NEWT.forward)."""
        film = self.td_mlp(emb, "newt.mlp")
        self._film = film
        film = F.interpolate(film, size=exciter.shape[-1], mode="linear")       # :69 (F.upsample)
        g_i, b_i, g_n, b_n = torch.split(film, self.n_waveshapers, 1)            # :70-72
        x = g_i * exciter + b_i                                                  # :74  dynamic.py:8
        self._lut_arg = x
        x = self.lut_shaper(x) if self.fast else self.exact_shaper(x)            # :75
        self._shaped = x
        x = g_n * x + b_n                                                        # :76
        return F.conv1d(x, self.w["newt.mixer.0.weight"], self.w["newt.mixer.0.bias"])  # :79

    # ---- noise ------------------------------------------------------------
    def fir_noise(self, H_re, noise):
        """models/modules/generators.py:21-35 (FIRNoiseSynth.forward)."""
        hop = self.control_hop
        H_z = torch.complex(H_re, torch.zeros_like(H_re))
        h = torch.fft.irfft(H_z.transpose(1, 2))
        h = h.roll(self.ir_length // 2, -1)
        h = h * self.window.view(1, 1, -1)
        H = torch.fft.rfft(h)
        X = torch.stft(noise, self.ir_length, hop, return_complex=True).unsqueeze(0)
        Y = X * H.transpose(1, 2)
        y = torch.istft(Y, self.ir_length, hop, center=False)
        return y.unsqueeze(1)[:, :, : H_re.shape[-1] * hop]

    # ---- reverb -----------------------------------------------------------
    def reverb(self, x):
        """models/modules/shaping.py:161-173 (Reverb.forward): circular convolution, length max(N, 32000)."""
        ir_ = torch.cat((self.w.get("reverb.initial_zero", torch.zeros(1, 1)), self.w["reverb.ir"]), dim=-1)
        if x.shape[-1] > ir_.shape[-1]:
            ir_ = F.pad(ir_, (0, x.shape[-1] - ir_.shape[-1]))
            x_ = x
        else:
            x_ = F.pad(x, (0, ir_.shape[-1] - x.shape[-1]))
        return x + torch.fft.irfft(torch.fft.rfft(x_) * torch.fft.rfft(ir_))[..., : x.shape[-1]]

    # ---- whole forward ------------------------------------------------------
    @torch.no_grad()
    def forward(self, f0, control, phase_u=None, noise=None, stages=None):
        """models/neural_waveshaping.py:74-90.  ``stages``: optional dict filled with intermediates."""
        f0, control = _t(f0).float(), _t(control).float()
        T = f0.shape[-1]
        N = T * self.control_hop
        f0_up = F.interpolate(f0, size=N, mode="linear")                         # :75 (F.upsample)
        if phase_u is None:
            phase_u = torch.rand_like(self.rand_phase)                            # RNG draw #1, generators.py:55
        phase_u = _t(phase_u).float().reshape(-1)
        x = self.exciter(f0_up, phase_u)                                          # :76
        emb = self.embedding(control)                                             # :78
        y_newt = self.newt(x, emb)                                                # :80
        H = self.td_mlp(emb, "h_generator")                                       # :82
        if noise is None:
            noise = torch.rand(self.control_hop * T - 1)                          # RNG draw #2, generators.py:30
        noise = _t(noise).float()
        y_noise = self.fir_noise(H, noise)                                        # :83
        pre = torch.cat((y_newt, y_noise), dim=1).sum(1)                          # :85-86
        y = self.reverb(pre)                                                      # :88
        if stages is not None:
            stages.update(f0_up=f0_up[:, 0], phase=self._phase, osc=self._osc, exciter=x, gru_out=self._gru_out,
                          embedding=emb, film=self._film, lut_arg=self._lut_arg, shaped=self._shaped,
                          newt_out=y_newt[:, 0], H=H, noise_out=y_noise[:, 0], pre_reverb=pre, y=y,
                          phase_u=phase_u, noise=noise)
        return y

    __call__ = forward


def load_weights_npz(path):
    """Weights fixture written by tests/golden/make_golden.py (flat arrays, no pickled classes)."""
    z = np.load(path)
    return {k: z[k] for k in z.files}
