"""Frame-rate building blocks (reference: models/modules/dynamic.py:6-40).

State-dict keys match the reference (``net.{0,3,6,9}.weight|bias``, ``net.{1,4,7}.layer_norm.weight|bias``) so its
checkpoints load unchanged.  Inside ``NeuralWaveshaping.forward`` this arithmetic (1x1 conv -> LayerNorm over channels ->
LeakyReLU(0.01), FiLM) runs fused in ``frame_mlps16_kernel`` / ``exciter_newt_kernel``; called on their own the modules
run the stand-alone stage kernels of csrc/stages.hip (any layer sizes), one launch per call.
"""
import torch
import torch.nn as nn

from ... import ginlite as gin
from . import _standalone as sa


class FiLM(nn.Module):
    """gamma * x + beta (reference dynamic.py:6-8) on CUDA tensors; operands are broadcast like the reference's."""

    def forward(self, x, gamma, beta):
        x, gamma, beta = torch.broadcast_tensors(x, gamma, beta)
        x, gamma, beta = sa.contiguous(x, "x"), sa.contiguous(gamma, "gamma"), sa.contiguous(beta, "beta")

        def c_call(L):
            y = torch.empty_like(x)
            with torch.cuda.device(x.device):
                sa.checked(L.nws_film(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), x.numel(), y.data_ptr(),
                                      sa.stream_ptr(x.device)), "nws_film")
            return y

        return sa.call("film", "nws_film", (x, gamma, beta), c_call)


class Conv1x1(nn.Conv1d):
    """nn.Conv1d(in, out, 1) with the reference's parameters / state-dict keys whose forward is a HIP kernel (g_conv1x1_kernel,
    csrc/generic.hip): harmonic_mixer (models/neural_waveshaping.py:54) and newt.mixer (shaping.py:63-65) called on their own -
    what the forward runs instead of the fused kernels when somebody has registered forward hooks on sub-modules."""

    def __init__(self, in_channels, out_channels, kernel_size=1, **kw):
        if kernel_size not in (1, (1,)) or kw.get("groups", 1) != 1:
            raise RuntimeError("Conv1x1: kernel_size 1, groups 1")
        super().__init__(in_channels, out_channels, 1, **kw)

    def forward(self, x):
        x = sa.contiguous(x, "x")
        if x.dim() != 3 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"Conv1d({self.in_channels}, {self.out_channels}, 1): expected (B, {self.in_channels}, N), got {tuple(x.shape)}")
        wt = sa._req(self.weight.detach(), "weight")
        bt = sa._req(self.bias.detach(), "bias") if self.bias is not None else None
        sa.no_autograd(params=[self.weight])
        B, Cin, N = x.shape

        def c_call(L):
            with torch.cuda.device(x.device):
                y = torch.empty((B, self.out_channels, N), dtype=torch.float32, device=x.device)
                sa.checked(L.nws_g_conv1x1(x.data_ptr(), wt.data_ptr(), bt.data_ptr() if bt is not None else None, B, Cin,
                                           self.out_channels, N, y.data_ptr(), sa.stream_ptr(x.device)), "nws_g_conv1x1")
            return y

        return sa.call("g_conv1x1", "nws_g_conv1x1", (x, wt, bt), c_call)


def upsample_linear(x, hop: int):
    """F.upsample(x, T * hop, mode="linear") on the last axis of a CUDA tensor (neural_waveshaping.py:75, shaping.py:69)."""
    x = sa.contiguous(x, "x")
    T = x.shape[-1]

    def c_call(L):
        with torch.cuda.device(x.device):
            y = torch.empty(tuple(x.shape[:-1]) + (T * hop,), dtype=torch.float32, device=x.device)
            sa.checked(L.nws_g_upsample(x.data_ptr(), x.numel() // T, T, int(hop), y.data_ptr(), sa.stream_ptr(x.device)), "nws_g_upsample")
        return y

    return sa.call("g_upsample", "nws_g_upsample", (x, int(hop)), c_call)


class TimeDistributedLayerNorm(nn.Module):
    """LayerNorm over the channel axis of a (B, C, T) tensor (reference dynamic.py:11-17)."""

    def __init__(self, size: int):
        super().__init__()
        self.layer_norm = nn.LayerNorm(size)

    def forward(self, x):
        x = sa.contiguous(x, "x")
        ln = self.layer_norm
        if x.dim() != 3 or x.shape[1] != ln.weight.numel():
            raise RuntimeError(f"TimeDistributedLayerNorm({ln.weight.numel()}): expected (B, {ln.weight.numel()}, T), got {tuple(x.shape)}")
        g, b = sa._req(ln.weight.detach(), "layer_norm.weight"), sa._req(ln.bias.detach(), "layer_norm.bias")

        def c_call(L):
            y = torch.empty_like(x)
            with torch.cuda.device(x.device):
                sa.checked(L.nws_td_layer_norm(x.data_ptr(), g.data_ptr(), b.data_ptr(), x.shape[0], x.shape[1], x.shape[2],
                                               float(ln.eps), y.data_ptr(), sa.stream_ptr(x.device)), "nws_td_layer_norm")
            return y

        return sa.call("td_layer_norm", "nws_td_layer_norm", (x, g, b, float(ln.eps)), c_call)


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
        """(B, in_size, T) -> (B, out_size, T) on a CUDA tensor: one launch of td_mlp_kernel (csrc/stages.hip)."""
        x = sa.contiguous(x, "x")
        if x.dim() != 3 or x.shape[1] != self.in_size:
            raise RuntimeError(f"TimeDistributedMLP: expected (B, {self.in_size}, T), got {tuple(x.shape)}")
        return td_mlp_forward(x, self.net)


def td_mlp_forward(x, net):
    """Run a Conv1d(k=1) [-> TimeDistributedLayerNorm -> LeakyReLU] ... stack (or a bare Conv1d) on (B, C, T)."""
    mods = list(net) if isinstance(net, (nn.Sequential, list, tuple)) else [net]
    convs = [m for m in mods if isinstance(m, nn.Conv1d)]
    norms = [m for m in mods if isinstance(m, TimeDistributedLayerNorm)]
    acts = [m for m in mods if isinstance(m, nn.LeakyReLU)]
    if len(norms) != len(convs) - 1 or any(c.kernel_size != (1,) or c.groups != 1 for c in convs):
        raise RuntimeError("TimeDistributedMLP: unexpected layer stack")
    ws = [sa._req(c.weight.detach(), "net weight") for c in convs]
    bs = [sa._req(c.bias.detach(), "net bias") for c in convs]
    gs = [sa._req(n.layer_norm.weight.detach(), "layer_norm.weight") for n in norms]
    ls = [sa._req(n.layer_norm.bias.detach(), "layer_norm.bias") for n in norms]
    eps = float(norms[0].layer_norm.eps) if norms else 1e-5
    slope = float(acts[0].negative_slope) if acts else 0.01
    # one eps / slope per launch (nws_td_mlp): a stack that mixes them is not what the kernel computes
    if any(float(n.layer_norm.eps) != eps for n in norms) or any(float(a.negative_slope) != slope for a in acts):
        raise RuntimeError("TimeDistributedMLP: the stage kernel takes ONE LayerNorm eps and ONE LeakyReLU slope for the whole "
                           "stack; these layers differ")
    if len(acts) != len(norms) or any(not n.layer_norm.elementwise_affine for n in norms):
        raise RuntimeError("TimeDistributedMLP: expected Conv1d -> LayerNorm(affine) -> LeakyReLU blocks")
    sa.no_autograd(params=[p for c in convs for p in (c.weight, c.bias)])
    depth = len(convs)
    hidden = ws[0].shape[0]
    out_size = ws[-1].shape[0]

    def c_call(L):
        arr = lambda ts: (sa.C.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts])  # noqa: E731
        y = torch.empty((x.shape[0], out_size, x.shape[2]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            sa.checked(L.nws_td_mlp(x.data_ptr(), x.shape[0], x.shape[1], hidden, out_size, depth, x.shape[2], arr(ws), arr(bs),
                                    arr(gs), arr(ls), eps, slope, y.data_ptr(), sa.stream_ptr(x.device)), "nws_td_mlp")
        return y

    return sa.call("td_mlp", "nws_td_mlp", (x, ws, bs, gs, ls, eps, slope), c_call)
