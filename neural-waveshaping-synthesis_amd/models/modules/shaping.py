"""Waveshaper bank and reverb (reference: models/modules/shaping.py:10-173).

Constructor signatures, attribute names and state-dict keys follow the reference so that
``model.newt = FastNEWT(model.newt)`` (scripts/time_forward_pass.py:42-43, colab cell 17) and
checkpoint loading work unchanged.  The arithmetic lives in csrc/exciter_newt.hip (FiLM -> shaper
LUT / sin-MLP -> FiLM -> 64->1 mix, fused with the exciter) and csrc/reverb_fft.hip.
"""
import torch
import torch.nn as nn

from ... import ginlite as gin
from .dynamic import FiLM, TimeDistributedMLP
from ._fused import fused_only


class Sine(nn.Module):
    def forward(self, x):
        raise fused_only("Sine", "the shaper kernels (nws_sinf in csrc/nws_common.h)")


@gin.configurable
class TrainableNonlinearity(nn.Module):
    """64 independent scalar sin-MLPs stored as grouped 1x1 convs (reference shaping.py:15-37)."""

    def __init__(self, channels, width, nonlinearity=nn.ReLU, final_nonlinearity=Sine, depth=3):
        super().__init__()
        self.input_scale = nn.Parameter(torch.randn(1, channels, 1) * 10)
        stack = []
        for i in range(depth):
            last = i == depth - 1
            stack.append(nn.Conv1d(channels if i == 0 else channels * width,
                                   channels if last else channels * width, 1, groups=channels))
            stack.append(final_nonlinearity() if last else nonlinearity())
        self.net = nn.Sequential(*stack)
        self.channels, self.width, self.depth = channels, width, depth

    def forward(self, x):
        """Exact shapers on a (B, 64, N) CUDA tensor - one HIP kernel (nws_shaper_apply)."""
        from ...engine import _req  # local import: keep module import light
        from ... import _lib
        import ctypes as C

        x = _req(x, "x")
        if x.dim() != 3 or x.shape[1] != self.channels:
            raise RuntimeError(f"expected (B, {self.channels}, N), got {tuple(x.shape)}")
        w = _lib.NwsWeights()
        keep = []

        def P(t, n):
            t = _req(t.detach(), "shaping_fn", n)
            keep.append(t)
            return t.data_ptr()

        if self.depth != 4 or self.width != 8 or self.channels != 64:
            raise RuntimeError("kernels are specialised for 64 shapers, width 8, depth 4")
        w.shaper_in_scale = P(self.input_scale, 64)
        w.shaper_w0, w.shaper_b0 = P(self.net[0].weight, 512), P(self.net[0].bias, 512)
        w.shaper_w2, w.shaper_b2 = P(self.net[2].weight, 4096), P(self.net[2].bias, 512)
        w.shaper_w4, w.shaper_b4 = P(self.net[4].weight, 4096), P(self.net[4].bias, 512)
        w.shaper_w6, w.shaper_b6 = P(self.net[6].weight, 512), P(self.net[6].bias, 64)
        w.lut = None
        y = torch.empty_like(x)
        _lib.check(_lib.lib().nws_shaper_apply(C.byref(w), x.data_ptr(), x.shape[0], x.shape[2], y.data_ptr(),
                                               _lib.stream_ptr()), "nws_shaper_apply")
        return y


@gin.configurable
class NEWT(nn.Module):
    def __init__(self, n_waveshapers: int, control_embedding_size: int, shaping_fn_size: int = 16,
                 out_channels: int = 1):
        super().__init__()
        self.n_waveshapers = n_waveshapers
        self.mlp = TimeDistributedMLP(control_embedding_size, control_embedding_size, n_waveshapers * 4, depth=4)
        self.waveshaping_index = FiLM()
        self.shaping_fn = TrainableNonlinearity(n_waveshapers, shaping_fn_size, nonlinearity=Sine)
        self.normalising_coeff = FiLM()
        self.mixer = nn.Sequential(nn.Conv1d(n_waveshapers, out_channels, 1))

    def forward(self, exciter, control_embedding):
        raise fused_only("NEWT", "NeuralWaveshaping.forward (frame_mlps_kernel + exciter_newt_kernel)")


class FastNEWT(NEWT):
    """Lookup-table replacement of the shaper MLPs (reference shaping.py:82-151).

    ``lookup_table[s, i] = shaper_s(linspace(table_min, table_max, table_size)[i])`` is evaluated by the
    HIP kernel ``shaper_table_kernel``.  Like the reference it shares ``mlp``, the FiLM modules and
    ``mixer`` with the wrapped NEWT.  The wrapped NEWT's trained ``shaping_fn`` stays registered as a
    sub-module (the reference keeps a dangling re-initialised one there).
    """

    def __init__(self, newt: NEWT, table_size: int = 4096, table_min: float = -3.0, table_max: float = 3.0):
        super().__init__()
        self.table_size = table_size
        self.table_min = table_min
        self.table_max = table_max
        self.n_waveshapers = newt.n_waveshapers
        self.mlp = newt.mlp
        self.waveshaping_index = newt.waveshaping_index
        self.normalising_coeff = newt.normalising_coeff
        self.mixer = newt.mixer
        self._modules["shaping_fn"] = newt._modules["shaping_fn"]
        self.lookup_table = self._init_lookup_table(newt, table_size, self.n_waveshapers, table_min, table_max)

    @staticmethod
    def _init_lookup_table(newt, table_size, n_waveshapers, table_min, table_max):
        import ctypes as C
        from ... import _lib
        from ...engine import _req

        sh = newt._modules["shaping_fn"]
        home = sh.input_scale.device
        if home.type == "cuda":
            dev = home
        elif torch.cuda.is_available():
            # the reference builds FastNEWT before .to(device) (scripts/time_forward_pass.py:42-45);
            # the table is still computed by the HIP kernel, then parked next to the module
            dev = torch.device("cuda", torch.cuda.current_device())
        else:
            raise _lib.NwsError("FastNEWT needs an AMD GPU to evaluate its lookup table (no CPU fallback)")
        if sh.depth != 4 or sh.width != 8 or n_waveshapers != 64:
            raise RuntimeError("kernels are specialised for 64 shapers, width 8, depth 4")
        w = _lib.NwsWeights()
        keep = []

        def P(t, n):
            t = _req(t.detach().to(dev).contiguous(), "shaping_fn", n)
            keep.append(t)
            return t.data_ptr()

        w.shaper_in_scale = P(sh.input_scale, 64)
        w.shaper_w0, w.shaper_b0 = P(sh.net[0].weight, 512), P(sh.net[0].bias, 512)
        w.shaper_w2, w.shaper_b2 = P(sh.net[2].weight, 4096), P(sh.net[2].bias, 512)
        w.shaper_w4, w.shaper_b4 = P(sh.net[4].weight, 4096), P(sh.net[4].bias, 512)
        w.shaper_w6, w.shaper_b6 = P(sh.net[6].weight, 512), P(sh.net[6].bias, 64)
        table = torch.empty((n_waveshapers, table_size), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().nws_shaper_table(C.byref(w), int(table_size), float(table_min), float(table_max),
                                                   table.data_ptr(), _lib.stream_ptr()), "nws_shaper_table")
            torch.cuda.current_stream().synchronize()
        return nn.Parameter(table.to(home))

    def shaping_fn(self, x):
        """LUT lookup with the reference's index quirks, on a (B, 64, N) CUDA tensor (nws_shaper_apply)."""
        import ctypes as C
        from ... import _lib
        from ...engine import _req

        x = _req(x, "x")
        w = _lib.NwsWeights()
        table = _req(self.lookup_table.detach(), "lookup_table", 64 * self.table_size)
        w.lut, w.lut_size, w.lut_min, w.lut_max = table.data_ptr(), self.table_size, self.table_min, self.table_max
        y = torch.empty_like(x)
        _lib.check(_lib.lib().nws_shaper_apply(C.byref(w), x.data_ptr(), x.shape[0], x.shape[2], y.data_ptr(),
                                               _lib.stream_ptr()), "nws_shaper_apply")
        return y


@gin.configurable
class Reverb(nn.Module):
    """x + circular_conv(x, [0, ir]) of length max(N, len(ir)+1) (reference shaping.py:154-173)."""

    def __init__(self, length_in_seconds, sr):
        super().__init__()
        self.ir = nn.Parameter(torch.randn(1, sr * length_in_seconds - 1) * 1e-6)
        self.register_buffer("initial_zero", torch.zeros(1, 1))
        self._tables = {}

    def forward(self, x):
        """Stand-alone reverb on a (B, N) CUDA tensor: four-step FFT kernels of csrc/reverb_fft.hip."""
        import ctypes as C
        from ... import _lib
        from ...engine import _req, reverb_plan_and_tables

        x = _req(x, "x")
        if x.dim() != 2:
            raise RuntimeError(f"expected (B, N), got {tuple(x.shape)}")
        ir = _req(self.ir.detach(), "reverb.ir")
        B, N = x.shape
        plan, tables = reverb_plan_and_tables(x.device, N, ir.numel() + 1)
        L = _lib.lib()
        key = (plan.L, ir.data_ptr(), ir._version)
        spec = self._tables.get(key)
        if spec is None:
            self._tables.clear()
            spec = torch.empty(L.nws_reverb_spectrum_bytes(C.byref(plan)) // 4, dtype=torch.float32, device=x.device)
            nb = L.nws_reverb_workspace_bytes(C.byref(plan), 1)
            ws1 = torch.empty(nb, dtype=torch.uint8, device=x.device)
            _lib.check(L.nws_reverb_ir_spectrum(C.byref(plan), tables.data_ptr(), ir.data_ptr(), ir.numel(),
                                                spec.data_ptr(), ws1.data_ptr(), nb, _lib.stream_ptr()),
                       "nws_reverb_ir_spectrum")
            self._tables[key] = spec
        nb = L.nws_reverb_workspace_bytes(C.byref(plan), B)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        y = torch.empty_like(x)
        _lib.check(L.nws_reverb(C.byref(plan), tables.data_ptr(), spec.data_ptr(), x.data_ptr(), B, N, y.data_ptr(),
                                ws.data_ptr(), nb, _lib.stream_ptr()), "nws_reverb")
        return y
