"""Host-side worst-case error bounds that decide which arithmetic the fused kernels may use.

The reference contracts the 101 -> 64 harmonic mixer in fp32 (``nn.Conv1d``, models/neural_waveshaping.py:54,66).  The
fused oscillator kernel's default is a two-term fp16 split of BOTH operands (22-bit products, fp32 accumulation: fp32-class).
``NWS_EXCITER_HYBRID_W`` runs harmonics 16..n as plain fp16 x fp16 products instead; this module bounds what that costs
at the OUTPUT of ``NeuralWaveshaping.forward`` for ANY input, from the weights alone, in the style of
``Engine.fp16_mlp_safe``: nothing here looks at an input, an energy share or a typical case.

    |dx[s]|      <=  sum_{k >= 16} ( |W[s,k]| (2^-11 + 2^-12 (1 + 2^-11)) + subnormal term )         mixer input of shaper s
    |d idx[s]|   <=  G_idx[s] |dx[s]|                G_idx[s]  = max over inputs of |gamma_index[s]|    (FiLM, shaping.py:74)
    |d shaped|   <=  slope[s] |d idx[s]|             slope[s]  = max |T[s,i+1] - T[s,i]| size / range    (FastNEWT lerp, :149-150)
    |d newt|     <=  sum_s M[s] |d shaped[s]|        M[s]      = max over inputs of |gamma_norm[s]| |mixer.weight[s]|  (:76,:79)
    |d y|        <=  (1 + ||ir||_1) |d newt|         (Reverb, shaping.py:161-173: y = x + x * ir)

The FiLM gains are outputs of ``newt.mlp`` (TimeDistributedMLP, dynamic.py:20-40): its last layer sees a LayerNorm + LeakyReLU
output, bounded by sqrt(C - 1) |ln.weight| + |ln.bias| per channel whatever the input, so
max |gamma[c]| <= sum_j |W9[c, j]| lnb[j] + |b9[c]|.  Linear interpolation between frames (F.upsample) cannot exceed the
frame values.
"""
from __future__ import annotations

import math

import torch

U16 = 2.0 ** -11            # unit roundoff of fp16 (11-bit significand, round to nearest)
SUBNORMAL_HALF_ULP = 2.0 ** -25   # below 2^-14 fp16 spacing is 2^-24
K_SPLIT = 15                # harmonics 1..15 (+ the bias slot) keep the two-term split in the hybrid forms


def film_gain_bounds(mlp) -> torch.Tensor:
    """max over all inputs of |newt.mlp(x)[c]| per output channel c (see module docstring); `mlp` is a TimeDistributedMLP."""
    convs = [m for m in mlp.net if isinstance(m, torch.nn.Conv1d)]
    norms = [m for m in mlp.net if hasattr(m, "layer_norm")]
    last, ln = convs[-1], norms[-1].layer_norm
    C = ln.weight.numel()
    lnb = math.sqrt(max(C - 1, 1)) * ln.weight.detach().double().abs() + ln.bias.detach().double().abs()     # (C)
    w = last.weight.detach().double().abs().flatten(1)                                                     # (out, C)
    return w @ lnb + last.bias.detach().double().abs()


def hybrid_w_error_bound(model, table: torch.Tensor | None = None, table_min: float = -3.0, table_max: float = 3.0) -> dict:
    """Worst-case |y_hybrid_w - y_two_term| at the output of forward() for `model` (a NeuralWaveshaping with a FastNEWT).

    Returns {"bound": float, "dx_max": ..., "film_idx_max": ..., "slope_max": ..., "film_norm_mix_max": ..., "reverb_gain": ...}.
    `table` defaults to model.newt.lookup_table."""
    with torch.no_grad():
        W = model.harmonic_mixer.weight.detach().double().flatten(1).cpu()           # (S, n_harmonics)
        S = W.shape[0]
        hi = W[:, K_SPLIT:].abs()
        # rounding of the weight (times |sin| <= 1): relative 2^-11 in the normal range, half a subnormal ulp below 2^-14,
        # nothing for an exact zero
        per_w = torch.where(hi >= 2.0 ** -14, hi * U16, torch.where(hi > 0, torch.full_like(hi, SUBNORMAL_HALF_ULP),
                                                                    torch.zeros_like(hi)))
        per_s = hi * (U16 / 2) * (1 + U16)                                          # rounding of the sine (|sin| <= 1: half-ulp 2^-12)
        dx = (per_w + per_s).sum(1)                                                 # (S)
        g = film_gain_bounds(model.newt.mlp).cpu()                                  # (4 S): [g_idx | b_idx | g_norm | b_norm]
        g_idx, g_norm = g[:S], g[2 * S:3 * S]
        if table is None:
            table = model.newt.lookup_table
        t = table.detach().double().cpu()
        size = t.shape[1]
        slope = (t[:, 1:] - t[:, :-1]).abs().max(1).values * (size / (float(table_max) - float(table_min)))   # per unit of x
        mix = model.newt.mixer[0].weight.detach().double().abs().cpu().flatten(1).sum(0)                       # (S), out channels summed
        d_newt = float((g_norm * mix * slope * g_idx * dx).sum())
        rv = 1.0 + float(model.reverb.ir.detach().double().abs().sum())
        return {"bound": d_newt * rv, "dx_max": float(dx.max()), "film_idx_max": float(g_idx.max()),
                "slope_max": float(slope.max()), "film_norm_mix_max": float((g_norm * mix).max()), "reverb_gain": rv}
