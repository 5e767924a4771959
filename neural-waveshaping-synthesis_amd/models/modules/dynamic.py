"""Parameter containers for the frame-rate MLPs (reference: models/modules/dynamic.py:6-40).

State-dict keys match the reference (``net.{0,3,6,9}.weight|bias``,
``net.{1,4,7}.layer_norm.weight|bias``) so its checkpoints load unchanged.  The arithmetic
(1x1 conv -> LayerNorm over channels -> LeakyReLU(0.01)) runs inside the HIP kernel
``frame_mlps_kernel`` (csrc/frame_mlps.hip) on fp32 MFMA tiles.
"""
import torch.nn as nn

from ... import ginlite as gin
from ._fused import fused_only


class FiLM(nn.Module):
    """gamma * x + beta (reference dynamic.py:6-8); fused into csrc/exciter_newt.hip."""

    def forward(self, x, gamma, beta):
        raise fused_only("FiLM", "NeuralWaveshaping.forward (exciter_newt_kernel)")


class TimeDistributedLayerNorm(nn.Module):
    """LayerNorm over the channel axis of a (B, C, T) tensor (reference dynamic.py:11-17)."""

    def __init__(self, size: int):
        super().__init__()
        self.layer_norm = nn.LayerNorm(size)

    def forward(self, x):
        raise fused_only("TimeDistributedLayerNorm", "NeuralWaveshaping.forward (frame_mlps_kernel)")


@gin.configurable
class TimeDistributedMLP(nn.Module):
    def __init__(self, in_size: int, hidden_size: int, out_size: int, depth: int = 3):
        super().__init__()
        assert depth >= 3, "Depth must be at least 3"
        widths = [in_size] + [hidden_size] * (depth - 1) + [out_size]
        stack = []
        for i, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            stack.append(nn.Conv1d(fan_in, fan_out, 1))
            if i != depth - 1:
                stack += [TimeDistributedLayerNorm(hidden_size), nn.LeakyReLU()]
        self.net = nn.Sequential(*stack)
        self.in_size, self.hidden_size, self.out_size, self.depth = in_size, hidden_size, out_size, depth

    def forward(self, x):
        raise fused_only("TimeDistributedMLP", "NeuralWaveshaping.forward (frame_mlps_kernel)")
