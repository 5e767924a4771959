"""Signal generators of the NEWT synthesiser (reference: models/modules/generators.py:11-66).

These classes keep the reference's constructor signatures, buffers and state-dict keys; the DSP
itself runs in csrc/exciter_newt.hip (oscillator bank, fused with the 101->64 mixer and the
waveshapers) and csrc/fir_noise.hip (time-varying FIR noise).
"""
import math
from typing import Callable

import torch
import torch.nn as nn

from ... import ginlite as gin
from ._fused import fused_only


@gin.configurable
class FIRNoiseSynth(nn.Module):
    def __init__(self, ir_length: int, hop_length: int, window_fn: Callable = torch.hann_window):
        super().__init__()
        self.ir_length = ir_length
        self.hop_length = hop_length
        self.register_buffer("window", window_fn(ir_length))

    def forward(self, H_re):
        raise fused_only("FIRNoiseSynth", "NeuralWaveshaping.forward (frame_mlps_kernel + fir_noise_kernel)")


@gin.configurable
class HarmonicOscillator(nn.Module):
    def __init__(self, n_harmonics, sample_rate):
        super().__init__()
        self.sample_rate = sample_rate
        self.n_harmonics = n_harmonics
        # same buffers as the reference (generators.py:44-48): k = 1..n as int64, rand_phase = tau
        self.register_buffer("harmonic_axis", torch.arange(1, n_harmonics + 1).view(1, -1, 1))
        self.register_buffer("rand_phase", torch.full((1, n_harmonics, 1), math.tau))

    def forward(self, f0):
        raise fused_only("HarmonicOscillator", "NeuralWaveshaping.render_exciter / forward (exciter_newt_kernel)")
