"""Waveshaper bank and reverb (reference: models/modules/shaping.py:10-173).

Constructor signatures, attribute names and state-dict keys follow the reference so that
``model.newt = FastNEWT(model.newt)`` (scripts/time_forward_pass.py:42-43, colab cell 17) and
checkpoint loading work unchanged.  Inside ``NeuralWaveshaping.forward`` the arithmetic lives in csrc/exciter_newt.hip
(FiLM -> shaper LUT / sin-MLP -> FiLM -> 64->1 mix, fused with the exciter) and csrc/reverb_fft.hip; called on their own,
``NEWT.forward(exciter, control_embedding)``, ``TrainableNonlinearity.forward``, ``FastNEWT.shaping_fn``, ``Sine`` and
``Reverb.forward`` each run one stand-alone stage kernel.
"""
import ctypes as C

import torch
import torch.nn as nn

from ... import ginlite as gin
from . import _standalone as sa
from .dynamic import Conv1x1, FiLM, TimeDistributedMLP, upsample_linear


class Sine(nn.Module):
    def forward(self, x):
        """sin(x) on a CUDA tensor with the engine's full-range sine (nws_sin: <= 1.5e-7 absolute)."""
        x = sa.contiguous(x, "x")

        def c_call(L):
            y = torch.empty_like(x)
            with torch.cuda.device(x.device):
                sa.checked(L.nws_sin(x.data_ptr(), y.data_ptr(), x.numel(), sa.stream_ptr(x.device)), "nws_sin")
            return y

        return sa.call("sine", "nws_sin", (x,), c_call)


def _shaper_apply(x, wdesc_tuple):
    w, _, wdesc = wdesc_tuple

    def c_call(L):
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            sa.checked(L.nws_shaper_apply(C.byref(w), x.data_ptr(), x.shape[0], x.shape[2], y.data_ptr(), sa.stream_ptr(x.device)),
                       "nws_shaper_apply")
        return y

    return sa.call("shaper_apply", "nws_shaper_apply", (wdesc, x), c_call)


class _GShaperCache:
    """NwsShaperDesc (+ its byte tensor for the op layer) of a TrainableNonlinearity or a FastNEWT table, cached until a
    tensor changes"""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, sh=None, *, lut=None, lut_min=0.0, lut_max=0.0):
        from ...generic import desc_bytes, shaper_desc

        ts = [lut] if lut is not None else list(sh.parameters())
        key = tuple((t.data_ptr(), t._version) for t in ts) + (float(lut_min), float(lut_max))
        if key != self._key:
            sa.no_autograd(params=ts)
            keep = []
            d = shaper_desc(sh, lut=lut, lut_min=lut_min, lut_max=lut_max, keep=keep)
            self._val = (d, desc_bytes(d), keep)
            self._key = key
        return self._val[0], self._val[1]


def _g_shaper_apply(x, d, sdesc):
    def c_call(lib):
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            sa.checked(lib.nws_g_shaper_apply(C.byref(d), x.data_ptr(), x.shape[0] * x.shape[1], x.shape[2], y.data_ptr(),
                                              sa.stream_ptr(x.device)), "nws_g_shaper_apply")
        return y

    return sa.call("g_shaper_apply", "nws_g_shaper_apply", (sdesc, x), c_call)


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
        self._desc = sa.Desc()
        self._gdesc = _GShaperCache()

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_desc"] = sa.Desc()
        d["_gdesc"] = _GShaperCache()
        return d

    def forward(self, x):
        """Exact shapers on a (B, 64, N) CUDA tensor - one HIP kernel (nws_shaper_apply)."""
        x = sa.contiguous(x, "x")
        if x.dim() != 3 or x.shape[1] != self.channels:
            raise RuntimeError(f"expected (B, {self.channels}, N), got {tuple(x.shape)}")
        if not all(isinstance(m, Sine) for m in list(self.net)[1::2]):
            raise RuntimeError("kernels implement the sine activations NEWT configures (nonlinearity=Sine)")
        if (self.channels, self.width, self.depth) != (sa._lib.N_SHAPERS, sa._lib.SHAPER_WIDTH, 4):
            return _g_shaper_apply(x, *self._gdesc.get(self))          # any channels / width / depth (csrc/generic.hip)
        return _shaper_apply(x, self._desc.get(sa.shaper_fields(self)))


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
        self.mixer = nn.Sequential(Conv1x1(n_waveshapers, out_channels, 1))
        self._newt_desc = sa.Desc()
        self._g_newt = _GShaperCache()

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_newt_desc"] = sa.Desc()
        d["_g_newt"] = _GShaperCache()
        return d

    def _apply_fields(self):
        return sa.shaper_fields(self._modules["shaping_fn"]), {}

    def _specialised(self, hop):
        sh = self._modules["shaping_fn"]
        return (self.n_waveshapers == sa._lib.N_SHAPERS and self.mixer[0].out_channels == 1 and hop == sa._lib.HOP
                and (sh.channels, sh.width, sh.depth) == (sa._lib.N_SHAPERS, sa._lib.SHAPER_WIDTH, 4))

    def _g_shaper(self):
        return self._g_newt.get(self._modules["shaping_fn"])

    def _forward_generic(self, exciter, film):
        """any n_waveshapers / shaping_fn_size / depth / out_channels / hop: FiLM -> shaper -> FiLM (g_film_shaper_kernel) then
        the Conv1d(S -> out_channels) mixer (g_conv1x1_kernel), csrc/generic.hip"""
        d, sdesc = self._g_shaper()
        B, S, N = exciter.shape
        T = film.shape[2]
        mw = sa._req(self.mixer[0].weight.detach(), "newt.mixer.0.weight")
        mb = sa._req(self.mixer[0].bias.detach(), "newt.mixer.0.bias")
        O = mw.shape[0]

        def c_call(lib):
            with torch.cuda.device(exciter.device):
                shaped = torch.empty_like(exciter)
                out = torch.empty((B, O, N), dtype=torch.float32, device=exciter.device)
                st = sa.stream_ptr(exciter.device)
                sa.checked(lib.nws_g_film_shaper(C.byref(d), exciter.data_ptr(), film.data_ptr(), B, T, N // T, shaped.data_ptr(), st),
                           "nws_g_film_shaper")
                sa.checked(lib.nws_g_conv1x1(shaped.data_ptr(), mw.data_ptr(), mb.data_ptr(), B, S, O, N, out.data_ptr(), st),
                           "nws_g_conv1x1")
            return out

        return sa.call("g_newt_apply", "nws_g_film_shaper", (sdesc, exciter, film, mw, mb), c_call)

    def forward(self, exciter, control_embedding):
        """(B, 64, N) exciter, (B, 128, T) control embedding -> (B, 1, N) (reference shaping.py:67-79): the FiLM-parameter
        MLP (one launch) then FiLM -> shaper -> FiLM -> Conv1d(64 -> 1) on the materialised exciter (one launch)."""
        exciter = sa.contiguous(exciter, "exciter")
        if exciter.dim() != 3 or exciter.shape[1] != self.n_waveshapers:
            raise RuntimeError(f"NEWT: expected an exciter of shape (B, {self.n_waveshapers}, N), got {tuple(exciter.shape)}")
        film = self.mlp(control_embedding)                       # (B, 256, T), channel-major like the reference's Conv1d stack
        T = film.shape[2]
        if exciter.shape[0] != film.shape[0] or exciter.shape[2] % T:
            raise RuntimeError(f"NEWT: exciter {tuple(exciter.shape)} does not match {T} control frames")
        if sa.has_hooks(self.waveshaping_index, self.normalising_coeff, self.mixer, self.mixer[0], self._modules.get("shaping_fn")):
            # somebody listens on the inner modules: run them one by one, exactly the reference's sequence (shaping.py:69-79)
            fp = upsample_linear(film, exciter.shape[2] // T)
            g_i, b_i, g_n, b_n = torch.split(fp, self.n_waveshapers, 1)
            x = self.waveshaping_index(exciter, g_i, b_i)
            x = self.shaping_fn(x)
            x = self.normalising_coeff(x, g_n, b_n)
            return self.mixer(x)
        if not self._specialised(exciter.shape[2] // T):
            return self._forward_generic(exciter, film)
        tensors, scalars = self._apply_fields()
        tensors = dict(tensors, newt_out_w=self.mixer[0].weight, newt_out_b=self.mixer[0].bias)
        w, _, wdesc = self._newt_desc.get(tensors, scalars)

        def c_call(L):
            with torch.cuda.device(exciter.device):
                out = torch.empty((exciter.shape[0], 1, exciter.shape[2]), dtype=torch.float32, device=exciter.device)
                sa.checked(L.nws_newt_apply(C.byref(w), exciter.data_ptr(), film.data_ptr(), exciter.shape[0], T, out.data_ptr(),
                                            sa.stream_ptr(exciter.device)), "nws_newt_apply")
            return out

        return sa.call("newt_apply", "nws_newt_apply", (wdesc, exciter, film), c_call)


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
        self._lut_desc = sa.Desc()
        self._g_lut = _GShaperCache()

    def __getstate__(self):
        d = super().__getstate__()
        d["_lut_desc"] = sa.Desc()
        d["_g_lut"] = _GShaperCache()
        return d

    def _specialised(self, hop):
        return self.n_waveshapers == sa._lib.N_SHAPERS and self.mixer[0].out_channels == 1 and hop == sa._lib.HOP

    def _g_shaper(self):
        return self._g_lut.get(lut=self.lookup_table, lut_min=self.table_min, lut_max=self.table_max)

    @staticmethod
    def _init_lookup_table(newt, table_size, n_waveshapers, table_min, table_max):
        sh = newt._modules["shaping_fn"]
        home = sh.input_scale.device
        if home.type == "cuda":
            dev = home
        elif torch.cuda.is_available():
            # the reference builds FastNEWT before .to(device) (scripts/time_forward_pass.py:42-45);
            # the table is still computed by the HIP kernel, then parked next to the module
            dev = torch.device("cuda", torch.cuda.current_device())
        else:
            raise sa._lib.NwsError("FastNEWT needs an AMD GPU to evaluate its lookup table (no CPU fallback)")
        if (sh.channels, sh.width, sh.depth) != (sa._lib.N_SHAPERS, sa._lib.SHAPER_WIDTH, 4):
            # any channels / width / depth: runtime-size table kernel (csrc/generic.hip: g_shaper_table_kernel)
            import copy

            from ...generic import desc_bytes, shaper_desc

            sh_dev = sh if home == dev else copy.deepcopy(sh).to(dev)
            keep = []
            d = shaper_desc(sh_dev, keep=keep)

            def g_call(lib):
                with torch.cuda.device(dev):
                    table = torch.empty((n_waveshapers, table_size), dtype=torch.float32, device=dev)
                    sa.checked(lib.nws_g_shaper_table(C.byref(d), int(table_size), float(table_min), float(table_max), table.data_ptr(),
                                                      sa.stream_ptr(dev)), "nws_g_shaper_table")
                return table

            with torch.no_grad():
                table = sa.call("g_shaper_table", "nws_g_shaper_table",
                                (desc_bytes(d), keep[0], int(table_size), float(table_min), float(table_max)), g_call)
            torch.cuda.current_stream(dev).synchronize()
            return nn.Parameter(table.to(home))
        fields = {k: v.detach().to(dev).contiguous() for k, v in sa.shaper_fields(sh).items()}
        w, keep, wdesc = sa.Desc().get(fields)
        like = keep[0]

        def c_call(L):
            with torch.cuda.device(dev):
                table = torch.empty((n_waveshapers, table_size), dtype=torch.float32, device=dev)
                sa.checked(L.nws_shaper_table(C.byref(w), int(table_size), float(table_min), float(table_max), table.data_ptr(),
                                              sa.stream_ptr(dev)), "nws_shaper_table")
            return table

        table = sa.call("shaper_table", "nws_shaper_table", (wdesc, like, int(table_size), float(table_min), float(table_max)), c_call)
        torch.cuda.current_stream(dev).synchronize()
        return nn.Parameter(table.to(home))

    def _lut_fields(self):
        return {"lut": self.lookup_table}, {"lut_size": int(self.table_size), "lut_min": float(self.table_min),
                                            "lut_max": float(self.table_max)}

    def _apply_fields(self):
        return self._lut_fields()

    def shaping_fn(self, x):
        """LUT lookup with the reference's index quirks, on a (B, 64, N) CUDA tensor (nws_shaper_apply)."""
        x = sa.contiguous(x, "x")
        if x.dim() != 3 or x.shape[1] != self.n_waveshapers:
            raise RuntimeError(f"expected (B, {self.n_waveshapers}, N), got {tuple(x.shape)}")
        if self.lookup_table.numel() != self.n_waveshapers * self.table_size:
            raise RuntimeError("lookup_table does not match table_size")
        if self.n_waveshapers != sa._lib.N_SHAPERS:
            return _g_shaper_apply(x, *self._g_shaper())
        return _shaper_apply(x, self._lut_desc.get(*self._lut_fields()))


@gin.configurable
class Reverb(nn.Module):
    """x + circular_conv(x, [0, ir]) of length max(N, len(ir)+1) (reference shaping.py:154-173)."""

    def __init__(self, length_in_seconds, sr):
        super().__init__()
        self.ir = nn.Parameter(torch.randn(1, sr * length_in_seconds - 1) * 1e-6)
        self.register_buffer("initial_zero", torch.zeros(1, 1))
        self._tables = {}

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_tables"] = {}
        return d

    def forward(self, x):
        """Stand-alone reverb on a (B, N) CUDA tensor: four-step FFT kernels of csrc/reverb_fft.hip."""
        from ...engine import reverb_plan_and_tables

        x = sa.contiguous(x, "x")
        if x.dim() != 2:
            raise RuntimeError(f"expected (B, N), got {tuple(x.shape)}")
        ir = sa._req(self.ir.detach(), "reverb.ir")
        if ir.device != x.device:
            raise RuntimeError(f"x is on {x.device} but reverb.ir is on {ir.device}")
        B, N = x.shape
        from ..._lib import NwsReverbPlan

        probe = NwsReverbPlan()
        if sa._lib.lib().nws_reverb_plan(int(N), int(ir.numel()) + 1, C.byref(probe)) != 0:
            # odd circular length (every even one has a plan): the reference's own rfft / irfft expression (csrc/generic.hip)
            def d_call(lib):
                y = torch.empty_like(x)
                with torch.cuda.device(x.device):
                    sa.checked(lib.nws_g_reverb_direct(x.data_ptr(), ir.data_ptr(), ir.numel(), B, N, y.data_ptr(),
                                                       sa.stream_ptr(x.device)), "nws_g_reverb_direct")
                return y

            return sa.call("g_reverb_direct", "nws_g_reverb_direct", (x, ir.reshape(-1)), d_call)
        plan, tables, plan_t = reverb_plan_and_tables(x.device, N, ir.numel() + 1)
        L = sa._lib.lib()
        key = (plan.L, ir.data_ptr(), ir._version)
        spec = self._tables.get(key)
        if spec is None:
            self._tables.clear()
            with torch.cuda.device(x.device):
                spec = torch.empty(L.nws_reverb_spectrum_bytes(C.byref(plan)) // 4, dtype=torch.float32, device=x.device)
                nb = L.nws_reverb_workspace_bytes(C.byref(plan), 1)
                ws1 = torch.empty(nb, dtype=torch.uint8, device=x.device)
                sa.checked(L.nws_reverb_ir_spectrum(C.byref(plan), tables.data_ptr(), ir.data_ptr(), ir.numel(),
                                                    spec.data_ptr(), ws1.data_ptr(), nb, sa.stream_ptr(x.device)),
                           "nws_reverb_ir_spectrum")
            self._tables[key] = spec

        def c_call(L):
            with torch.cuda.device(x.device):
                nb = L.nws_reverb_workspace_bytes(C.byref(plan), B)
                ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
                y = torch.empty_like(x)
                sa.checked(L.nws_reverb(C.byref(plan), tables.data_ptr(), spec.data_ptr(), x.data_ptr(), B, N, y.data_ptr(),
                                        ws.data_ptr(), nb, sa.stream_ptr(x.device)), "nws_reverb")
            return y

        return sa.call("reverb", "nws_reverb", (plan_t, tables, spec, x), c_call)
