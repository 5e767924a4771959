"""Signal generators of the NEWT synthesiser (reference: models/modules/generators.py:11-66).

These classes keep the reference's constructor signatures, buffers and state-dict keys.  Inside
``NeuralWaveshaping.forward`` the DSP runs fused (csrc/exciter_newt.hip: oscillator bank + 101->64 mixer + waveshapers;
csrc/fir_noise.hip: time-varying FIR noise); called on their own the modules run the stand-alone stage kernels
(``oscillator_kernel``; ``fir_from_h_kernel`` + the noise kernel), drawing from the device's default generator exactly
like the reference (``rand_like`` for the phase offsets, ``rand(hop * T - 1)`` for the excitation).
"""
import math
from typing import Callable

import torch
import torch.nn as nn

from ... import ginlite as gin
from . import _standalone as sa


@gin.configurable
class FIRNoiseSynth(nn.Module):
    def __init__(self, ir_length: int, hop_length: int, window_fn: Callable = torch.hann_window):
        super().__init__()
        self.ir_length = ir_length
        self.hop_length = hop_length
        self.register_buffer("window", window_fn(ir_length))
        self._design = {}

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_design"] = {}
        return d

    def _design_matrix(self, dev):
        """window * roll(irfft(.), ir_length / 2) folded into one (256, 132) matrix (nws_fir_design_matrix), per window"""
        win = sa._req(self.window.detach(), "noise_synth.window", sa._lib.FIR_LEN)
        key = (win.data_ptr(), win._version)
        hit = self._design.get(key)
        if hit is None:
            self._design.clear()
            hit = torch.empty(sa._lib.FIR_LEN * sa._lib.FIR_DESIGN_COLS, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                sa.checked(sa._lib.lib().nws_fir_design_matrix(win.data_ptr(), hit.data_ptr(), sa.stream_ptr(dev)),
                           "nws_fir_design_matrix")
            self._design[key] = hit
        return hit

    def forward(self, H_re, *, noise=None):
        """H_re (B, ir_length/2 + 1, T) real filter magnitudes -> (B, 1, hop * T) filtered noise (generators.py:21-35).
        `noise` injects the excitation draw (hop * T - 1 samples) for parity tests."""
        if self.ir_length != sa._lib.FIR_LEN or self.hop_length != sa._lib.HOP or not self._window_symmetric():
            return self._forward_generic(H_re, noise)
        H = sa.contiguous(H_re, "H_re")
        if H.dim() != 3 or H.shape[1] != sa._lib.N_BANDS:
            raise RuntimeError(f"FIRNoiseSynth: expected (B, {sa._lib.N_BANDS}, T), got {tuple(H.shape)}")
        B, _, T = H.shape
        if T < 2:
            raise RuntimeError("need at least 2 frames (reflect padding of the noise STFT, generators.py:31)")
        D = self._design_matrix(H.device)
        if noise is None:
            noise = torch.rand(self.hop_length * T - 1, device=H.device)          # the reference's draw (generators.py:30)
        noise = sa._req(noise, "noise", self.hop_length * T - 1)

        def c_call(L):
            with torch.cuda.device(H.device):
                fir = torch.empty((B, T, sa._lib.FIR_HALF), dtype=torch.float32, device=H.device)
                out = torch.empty((B, T * sa._lib.HOP), dtype=torch.float32, device=H.device)
                sa.checked(L.nws_fir_from_h(H.data_ptr(), D.data_ptr(), B, T, fir.data_ptr(), sa.stream_ptr(H.device)), "nws_fir_from_h")
                sa.checked(L.nws_fir_noise(fir.data_ptr(), noise.data_ptr(), None, B, T, out.data_ptr(), sa.stream_ptr(H.device)),
                           "nws_fir_noise")
            return out

        o = sa.ops()
        out = o.fir_noise(o.fir_from_h(H, D), noise, None, -1) if o is not None else c_call(sa._lib.lib())
        return out.unsqueeze(1)


    def _window_symmetric(self):
        """the specialised kernels pass half rows of taps (include/nws_hip.h, nws_frame_mlps): needs a window that is symmetric
        about tap L/2 with window[0] == 0, like the reference's periodic Hann"""
        win = self.window
        key = (win.data_ptr(), win._version)
        hit = self.__dict__.get("_win_ok")
        if hit is None or hit[0] != key:
            w = win.detach().float().cpu()
            L = w.numel()
            top = float(w.abs().max())          # symmetric to fp32 rounding (torch.hann_window itself is 1.8e-7 off)
            hit = (key, bool(abs(float(w[0])) <= 1e-7 * top and float((w[1:L // 2] - w[L // 2 + 1:].flip(0)).abs().max()) <= 4e-7 * top))
            self.__dict__["_win_ok"] = hit
        return hit[1]

    def _forward_generic(self, H_re, noise):
        """any even ir_length >= hop_length: runtime-size stage kernels (csrc/generic.hip: g_fir_design_kernel, g_fir_noise_kernel)"""
        L, hop = int(self.ir_length), int(self.hop_length)
        H = sa.contiguous(H_re, "H_re")
        if H.dim() != 3 or H.shape[1] != L // 2 + 1:
            raise RuntimeError(f"FIRNoiseSynth: expected (B, {L // 2 + 1}, T), got {tuple(H.shape)}")
        if L % 2 or L < hop:
            raise RuntimeError(f"FIRNoiseSynth: ir_length {L} must be even and >= hop_length {hop} (torch.istft, generators.py:34)")
        B, _, T = H.shape
        win = sa._req(self.window.detach(), "noise_synth.window", L)
        if noise is None:
            noise = torch.rand(hop * T - 1, device=H.device)                      # the reference's draw (generators.py:30)
        noise = sa._req(noise, "noise", hop * T - 1)

        def c_call(lib):
            with torch.cuda.device(H.device):
                fir = torch.empty((B, T, L), dtype=torch.float32, device=H.device)
                out = torch.empty((B, T * hop), dtype=torch.float32, device=H.device)
                st = sa.stream_ptr(H.device)
                sa.checked(lib.nws_g_fir_design(H.data_ptr(), win.data_ptr(), L, B, T, fir.data_ptr(), st), "nws_g_fir_design")
                sa.checked(lib.nws_g_fir_noise(fir.data_ptr(), noise.data_ptr(), L, hop, B, T, None, 0, out.data_ptr(), st),
                           "nws_g_fir_noise")
            return out

        return sa.call("g_fir_noise", "nws_g_fir_noise", (H, win, noise, hop), c_call).unsqueeze(1)


@gin.configurable
class HarmonicOscillator(nn.Module):
    def __init__(self, n_harmonics, sample_rate):
        super().__init__()
        self.sample_rate = sample_rate
        self.n_harmonics = n_harmonics
        # same buffers as the reference (generators.py:44-48): k = 1..n as int64, rand_phase = tau
        self.register_buffer("harmonic_axis", torch.arange(1, n_harmonics + 1).view(1, -1, 1))
        self.register_buffer("rand_phase", torch.full((1, n_harmonics, 1), math.tau))

    def forward(self, f0, *, phase_u=None):
        """(B, N) upsampled F0 in Hz -> (B, n_harmonics, N): sin(k * tau * cumsum(f0) / sr + shift_k) * [k f0 < sr / 2]
        (generators.py:58-66; the cumulative sum is accumulated in float64 like torch's CPU cumsum).  N must be a multiple
        of 128.  `phase_u` injects the U[0,1) phase draw for parity tests."""
        f0 = sa.contiguous(f0, "f0")
        if f0.dim() != 2 or f0.shape[1] < 1:
            raise RuntimeError(f"HarmonicOscillator: expected (B, N), got {tuple(f0.shape)}")
        K = int(self.n_harmonics)
        rp = sa._req(self.rand_phase.detach().reshape(-1), "osc.rand_phase", K)
        u = torch.rand_like(self.rand_phase) if phase_u is None else phase_u            # the reference's draw (generators.py:55)
        u = sa._req(u.reshape(-1), "phase_u", K)
        B, N = f0.shape
        if K != sa._lib.N_HARMONICS or N % sa._lib.HOP:
            # any harmonic count / length: runtime-size stage kernels (csrc/generic.hip: g_phase_kernel, g_oscillator_kernel)
            def g_call(lib):
                with torch.cuda.device(f0.device):
                    phase = torch.empty_like(f0)
                    out = torch.empty((B, K, N), dtype=torch.float32, device=f0.device)
                    st = sa.stream_ptr(f0.device)
                    sa.checked(lib.nws_g_phase(None, f0.data_ptr(), B, N, 1, float(self.sample_rate), None, phase.data_ptr(), st),
                               "nws_g_phase")
                    sa.checked(lib.nws_g_oscillator(f0.data_ptr(), phase.data_ptr(), u.data_ptr(), rp.data_ptr(), K, B, N,
                                                    float(self.sample_rate), out.data_ptr(), st), "nws_g_oscillator")
                return out

            return sa.call("g_oscillator", "nws_g_oscillator", (f0, u, rp, float(self.sample_rate)), g_call)

        def c_call(L):
            with torch.cuda.device(f0.device):
                carry = torch.empty((B, N // 32), dtype=torch.float64, device=f0.device)
                out = torch.empty((B, sa._lib.N_HARMONICS, N), dtype=torch.float32, device=f0.device)
                st = sa.stream_ptr(f0.device)
                sa.checked(L.nws_phase_carry(None, f0.data_ptr(), B, N // sa._lib.HOP, carry.data_ptr(), st), "nws_phase_carry")
                sa.checked(L.nws_oscillator(f0.data_ptr(), carry.data_ptr(), u.data_ptr(), rp.data_ptr(), B, N,
                                            float(self.sample_rate), out.data_ptr(), st), "nws_oscillator")
            return out

        return sa.call("oscillator", "nws_oscillator", (f0, u, rp, float(self.sample_rate)), c_call)
