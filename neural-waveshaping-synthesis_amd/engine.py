"""Host-side engine: turns a NeuralWaveshaping module's parameters into the C-ABI's pointer struct,
owns the weight-independent tables (FIR design matrix, reverb DFT tables) and the cached IR spectrum,
and enqueues the HIP kernels on torch's current stream.

Two bindings of the same C-ABI (include/nws_hip.h):
  * `torch.ops.newt_hip.*` (csrc/torch_ops.cpp -> libnws_torch_ops.so): dispatcher-visible custom ops with TORCH_CHECK
    argument validation, device guard and current-stream lookup in C++ - the default;
  * ctypes on libnws_hip.so directly (`NWS_BACKEND=ctypes`): the torch-free binding a foreign host would write
    (INTEGRATION.md), kept as a second test path.
Both end in the same `extern "C"` launchers; there is no CPU or PyTorch fallback behind either.

PyTorch is used here only as plumbing: device memory (tensors), the current HIP stream and the
device RNG that the reference itself draws from inside forward().
"""
from __future__ import annotations

import ctypes as C
import itertools
import math
import os

import torch

from . import _lib
from ._lib import NwsForwardAux, NwsReverbPlan, NwsWeights, check, ptr


def _req(t: torch.Tensor, name: str, numel: int | None = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if t.device.type != "cuda":
        raise _lib.NwsError(
            f"{name} lives on {t.device}: the NEWT forward path only runs as HIP kernels on an AMD GPU "
            "(move the model and inputs with .to('cuda')); there is no CPU fallback.")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    if numel is not None and t.numel() != numel:
        raise RuntimeError(f"{name}: expected {numel} elements, got {tuple(t.shape)} "
                           "(kernels are specialised for gin/models/newt.gin)")
    return t


def same_device(ref: torch.device, **tensors):
    """Every tensor of one call must live on the GPU the weights live on (the reference raises a RuntimeError for mixed
    devices as well; raw pointers would be mixed silently otherwise)."""
    for name, t in tensors.items():
        if t is not None and t.device != ref:
            raise RuntimeError(f"{name} is on {t.device} but the model's parameters are on {ref}: move them to the same GPU")


# ---- which binding ------------------------------------------------------------------------------------------------------
_OPS = None
_OPS_TRIED = False


def ops():
    """torch.ops.newt_hip, or None when the ctypes binding is selected (NWS_BACKEND=ctypes)."""
    global _OPS, _OPS_TRIED
    if not _OPS_TRIED:
        want = os.environ.get("NWS_BACKEND", "ops")
        if want not in ("ops", "ctypes"):
            raise _lib.NwsError(f"NWS_BACKEND={want!r}: expected 'ops' or 'ctypes'")
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnws_torch_ops.so")
        if want == "ops":
            if not os.path.exists(path):
                raise _lib.NwsError(f"{path} not found: build it (python __graft_entry__.py build) or select the ctypes "
                                    "binding with NWS_BACKEND=ctypes")
            _lib.lib()                                    # ABI / struct-layout checks of libnws_hip.so first
            torch.ops.load_library(path)
            if int(torch.ops.newt_hip.abi_version()) != _lib.ABI_VERSION:
                raise _lib.NwsError("libnws_torch_ops.so was built against another ABI version: rebuild")
            _OPS = torch.ops.newt_hip
        _OPS_TRIED = True
    return _OPS


def stream_ptr(dev: torch.device):
    return torch.cuda.current_stream(dev).cuda_stream


# ---- registry epoch: bumped whenever ANY module registers a parameter / buffer / sub-module ------------------------------
_EPOCH = [0]


def _bump(*_a, **_k):
    _EPOCH[0] += 1


try:   # torch >= 2.0
    from torch.nn.modules.module import (register_module_buffer_registration_hook, register_module_module_registration_hook,
                                         register_module_parameter_registration_hook)
    register_module_parameter_registration_hook(_bump)
    register_module_buffer_registration_hook(_bump)
    register_module_module_registration_hook(_bump)
except Exception:   # pragma: no cover
    pass


_TABLE_CACHE: dict = {}  # (device index, L, N1, N2) -> tables tensor (twiddles / DFT matrices of one transform length)
_PLAN_CACHE: dict = {}   # (device index, n_samples, ir_len_plus1) -> (plan, tables tensor, plan tensor)


def reverb_plan_and_tables(device: torch.device, n_samples: int, ir_len_plus1: int):
    """The reverb plan for a clip of n_samples (every even circular length has one: direct four-step transform or overlap-save
    blocks, csrc/reverb_fft.hip), the constant tables of its transform length (shared by all plans of that length) and the
    plan as the CPU int32 tensor the torch ops take."""
    pkey = (device.index, int(n_samples), int(ir_len_plus1))
    hit = _PLAN_CACHE.get(pkey)
    if hit is not None:
        return hit
    plan = NwsReverbPlan()
    check(_lib.lib().nws_reverb_plan(int(n_samples), int(ir_len_plus1), C.byref(plan)), "nws_reverb_plan")
    key = (device.index, plan.L, plan.N1, plan.N2)
    tables = _TABLE_CACHE.get(key)
    if tables is None:
        nbytes = _lib.lib().nws_reverb_table_bytes(C.byref(plan))
        with torch.cuda.device(device):
            tables = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
            check(_lib.lib().nws_reverb_build_tables(C.byref(plan), ptr(tables), stream_ptr(device)), "nws_reverb_build_tables")
            # one-time: make the tables visible to every stream before anybody can use them (callers may issue forwards
            # round-robin on several streams; the builder stream is whichever one got here first)
            torch.cuda.current_stream(device).synchronize()
        _TABLE_CACHE[key] = tables
    hit = (plan, tables, plan.as_tensor())
    if len(_PLAN_CACHE) >= 256:
        _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
    _PLAN_CACHE[pkey] = hit
    return hit


class Engine:
    """Per-model launcher.  ``model`` is a NeuralWaveshaping module (see models/neural_waveshaping.py)."""

    def __init__(self, model):
        self._model_ref = model
        self._w = None          # (NwsWeights, keep-alive list, device, wdesc tensor)
        self._fp = None         # what the cache was built from: ((data_ptr, version) per tensor, registry epoch, options)
        self._tensors = None    # the parameter / buffer objects the fingerprint walks
        self._fir_design = None
        self._spectra = {}      # L -> spectrum tensor
        self._workspaces = {}   # (B, T, stream) -> tensor

    # a copied / unpickled module gets a fresh engine (the caches hold raw device pointers of the ORIGINAL's tensors)
    def __deepcopy__(self, memo):
        return Engine(memo.get(id(self._model_ref), self._model_ref))

    def __reduce__(self):
        return (Engine, (self._model_ref,))

    # ---- which kernels: the fused ones are compiled for gin/models/newt.gin, everything else takes the runtime-size path ----
    def specialised(self) -> bool:
        """True when the module tree has the architecture the fused kernels are compiled for (101 harmonics, 64 shapers of
        width 8 / depth 4, GRU(2 -> 128), 128-d embedding, depth-4 frame MLPs, 256-tap FIR, hop 128, one output channel);
        any other gin configuration runs through generic.GenericEngine (csrc/generic.hip)."""
        hit = getattr(self, "_spec", None)
        if hit is not None and hit[0] == _EPOCH[0]:
            return hit[1] and self._window_symmetric()
        m = self._model_ref

        def mlp_ok(mlp, out_rows):
            convs = [c for c in mlp.net if isinstance(c, torch.nn.Conv1d)]
            return (len(mlp.net) == 10 and len(convs) == 4 and all(c.in_channels == _lib.HIDDEN for c in convs)
                    and all(c.out_channels == _lib.HIDDEN for c in convs[:3]) and convs[3].out_channels == out_rows)

        try:
            g = m.embedding.gru
            sh = m.newt._modules.get("shaping_fn")
            lut = getattr(m.newt, "lookup_table", None)
            ok = (m.control_hop == _lib.HOP and m.osc.n_harmonics == _lib.N_HARMONICS
                  and m.harmonic_mixer.out_channels == _lib.N_SHAPERS and m.harmonic_mixer.in_channels == _lib.N_HARMONICS
                  and g.input_size == 2 and g.hidden_size == _lib.HIDDEN and g.num_layers == 1 and not g.bidirectional
                  and m.embedding.proj.out_channels == _lib.HIDDEN and m.embedding.proj.in_channels == _lib.HIDDEN
                  and mlp_ok(m.newt.mlp, _lib.FILM_CH) and mlp_ok(m.h_generator, _lib.N_BANDS)
                  and m.noise_synth.ir_length == _lib.FIR_LEN and m.noise_synth.hop_length == _lib.HOP
                  and m.newt.n_waveshapers == _lib.N_SHAPERS and m.newt.mixer[0].out_channels == 1
                  and (lut is not None and lut.shape[0] == _lib.N_SHAPERS
                       or lut is None and sh is not None and sh.channels == _lib.N_SHAPERS and sh.width == 8 and sh.depth == 4))
        except AttributeError:
            ok = False
        self._spec = (_EPOCH[0], bool(ok))
        return bool(ok) and self._window_symmetric()

    def _window_symmetric(self) -> bool:
        """The fused kernels hand HALF of every frame's FIR taps from the frame MLPs to the noise kernel (include/nws_hip.h,
        nws_frame_mlps): valid when noise_synth.window is symmetric about tap L/2 with window[0] == 0 - the reference's
        periodic Hann (generators.py:20).  Any other window_fn takes the runtime-size path (full rows).  One 1 KB read per
        window version."""
        win = self._model_ref.noise_synth.window
        key = (win.data_ptr(), win._version)
        hit = self.__dict__.get("_win_ok")
        if hit is None or hit[0] != key:
            w = win.detach().float().cpu()
            L = w.numel()
            # (symmetric to fp32 rounding: torch.hann_window(256) itself differs by 1.8e-7 between taps 128 - d and 128 + d;
            # the kernels use the upper half's values for both, an error of that size on the window)
            ok = bool(L == _lib.FIR_LEN and abs(float(w[0])) <= 1e-7 * float(w.abs().max())
                      and float((w[1:L // 2] - w[L // 2 + 1:].flip(0)).abs().max()) <= 4e-7 * float(w.abs().max()))
            hit = self._win_ok = (key, ok)
        return hit[1]

    @property
    def generic(self):
        g = self.__dict__.get("_generic")
        if g is None:
            from .generic import GenericEngine

            g = self._generic = GenericEngine(self._model_ref)
        return g

    # ---- cache control -----------------------------------------------------------------------
    def invalidate(self):
        if self.__dict__.get("_generic") is not None:
            self._generic.invalidate()
        self._w = None
        self._fp = None
        self._tensors = None
        self._fir_design = None
        self._spectra.clear()
        self._workspaces.clear()

    def _fingerprint(self):
        """Cheap identity of everything the cached pointer struct and derived tables were computed from: the storage address
        and in-place version counter of every parameter and buffer (in-place updates - optimizer steps, `p.mul_()`,
        `p.copy_()`, `load_state_dict` on a sub-module - bump the version; re-homing changes the address), the module
        registry epoch (a replaced Parameter / sub-module object) and the kernel options.  ~12 us, once per forward.
        What it cannot see: writes through `p.data` (that view has its own version counter) - call
        `model.invalidate_cache()` after those."""
        m = self._model_ref
        if self._tensors is None or self._tensors[0] != _EPOCH[0]:
            self._tensors = (_EPOCH[0], list(itertools.chain(m.parameters(), m.buffers())))
        return (tuple((t.data_ptr(), t._version) for t in self._tensors[1]), id(m.newt), getattr(m, "exciter_opts", None),
                os.environ.get("NWS_EXCITER_OPTS"))

    # ---- weights -----------------------------------------------------------------------------
    def weights(self):
        """(NwsWeights struct, keep-alive list, device) - rebuilt when the parameters' fingerprint changed"""
        return self._wd()[:3]

    def _wd(self):
        fp = self._fingerprint()
        if self._w is not None and fp == self._fp:
            return self._w
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # building the pointer struct launches table kernels and synchronises: not legal inside a HIP-graph capture
            raise _lib.NwsError("the model's parameters / options changed (or were never staged) and the engine's derived tables "
                                "must be rebuilt, which cannot happen during HIP-graph capture: run one forward (or "
                                "model._engine.weights()) before capturing")
        if self._w is not None:
            # Other streams (ForwardPipeline, NewtStream, the caller's own) may still have kernels enqueued that read the
            # fragment tables / spectra / workspaces dropped below; the caching allocator only orders re-use on the stream a
            # block was allocated on.  A rebuild is rare (weights changed): drain the device first.
            torch.cuda.synchronize(self._w[2])
            self.invalidate()
            fp = self._fingerprint()
        m = self._model_ref
        keep = []

        def P(t, name, numel):
            t = _req(t.detach(), name, numel)
            keep.append(t)
            return t.data_ptr()

        w = NwsWeights()
        g = m.embedding.gru
        w.gru_w_ih = P(g.weight_ih_l0, "embedding.gru.weight_ih_l0", 384 * 2)
        w.gru_w_hh = P(g.weight_hh_l0, "embedding.gru.weight_hh_l0", 384 * 128)
        w.gru_b_ih = P(g.bias_ih_l0, "embedding.gru.bias_ih_l0", 384)
        w.gru_b_hh = P(g.bias_hh_l0, "embedding.gru.bias_hh_l0", 384)
        w.proj_w = P(m.embedding.proj.weight, "embedding.proj.weight", 128 * 128)
        w.proj_b = P(m.embedding.proj.bias, "embedding.proj.bias", 128)
        w.mixer_w = P(m.harmonic_mixer.weight, "harmonic_mixer.weight", 64 * 101)
        w.mixer_b = P(m.harmonic_mixer.bias, "harmonic_mixer.bias", 64)
        dev = keep[-1].device
        L = _lib.lib()
        with torch.cuda.device(dev):
            st = stream_ptr(dev)
            frags = torch.empty(28672, dtype=torch.uint8, device=dev)
            check(L.nws_mixer_frags(w.mixer_w, w.mixer_b, ptr(frags), st), "nws_mixer_frags")
            keep.append(frags)
            w.mixer_frags = frags.data_ptr()
            if not os.environ.get("NWS_EXCITER_NO_RANGE"):     # (A/B switch: every table lookup in the clamped form)
                xb = torch.empty(_lib.N_SHAPERS, dtype=torch.float32, device=dev)
                check(L.nws_exciter_bound(w.mixer_w, w.mixer_b, ptr(xb), st), "nws_exciter_bound")
                keep.append(xb)
                w.exciter_bound = xb.data_ptr()
            for name, mlp, out_rows, wf, bf, gf, lf in (
                    ("newt.mlp", m.newt.mlp, 256, w.newt_mlp_w, w.newt_mlp_b, w.newt_ln_g, w.newt_ln_b),
                    ("h_generator", m.h_generator, 129, w.hgen_w, w.hgen_b, w.hgen_ln_g, w.hgen_ln_b)):
                if len(mlp.net) != 10:
                    raise RuntimeError(f"{name}: kernels are specialised for depth=4 TimeDistributedMLP")
                for i in range(4):
                    rows = 128 if i < 3 else out_rows
                    wf[i] = P(mlp.net[3 * i].weight, f"{name}.net.{3 * i}.weight", rows * 128)
                    bf[i] = P(mlp.net[3 * i].bias, f"{name}.net.{3 * i}.bias", rows)
                    if i < 3:
                        gf[i] = P(mlp.net[3 * i + 1].layer_norm.weight, f"{name}.net.{3 * i + 1}.layer_norm.weight", 128)
                        lf[i] = P(mlp.net[3 * i + 1].layer_norm.bias, f"{name}.net.{3 * i + 1}.layer_norm.bias", 128)
            sh = m.newt._modules.get("shaping_fn")
            if sh is not None:
                if len(sh.net) != 8:
                    raise RuntimeError("newt.shaping_fn: kernels are specialised for depth=4, width=8")
                w.shaper_in_scale = P(sh.input_scale, "newt.shaping_fn.input_scale", 64)
                w.shaper_w0 = P(sh.net[0].weight, "newt.shaping_fn.net.0.weight", 512)
                w.shaper_b0 = P(sh.net[0].bias, "newt.shaping_fn.net.0.bias", 512)
                w.shaper_w2 = P(sh.net[2].weight, "newt.shaping_fn.net.2.weight", 4096)
                w.shaper_b2 = P(sh.net[2].bias, "newt.shaping_fn.net.2.bias", 512)
                w.shaper_w4 = P(sh.net[4].weight, "newt.shaping_fn.net.4.weight", 4096)
                w.shaper_b4 = P(sh.net[4].bias, "newt.shaping_fn.net.4.bias", 512)
                w.shaper_w6 = P(sh.net[6].weight, "newt.shaping_fn.net.6.weight", 512)
                w.shaper_b6 = P(sh.net[6].bias, "newt.shaping_fn.net.6.bias", 64)
                turns = torch.empty((64, _lib.SHAPER_TURNS_ROW), dtype=torch.float32, device=dev)
                check(L.nws_shaper_turns(C.byref(w), ptr(turns), st), "nws_shaper_turns")
                keep.append(turns)
                w.shaper_turns = turns.data_ptr()
            table = getattr(m.newt, "lookup_table", None)
            if table is not None:
                size = int(m.newt.table_size)
                w.lut = P(table, "newt.lookup_table", 64 * size)
                pairs = torch.empty((64, size, 2), dtype=torch.float32, device=table.device)
                check(L.nws_lut_pairs(w.lut, size, ptr(pairs), st), "nws_lut_pairs")
                keep.append(pairs)
                w.lut_pairs = pairs.data_ptr()
                w.lut_size = size
                w.lut_min = float(m.newt.table_min)
                w.lut_max = float(m.newt.table_max)
            else:
                w.lut = None
                w.lut_pairs = None
                w.lut_size, w.lut_min, w.lut_max = 0, 0.0, 0.0
            w.exciter_opts = self.exciter_opts()
            if getattr(m.newt, "lookup_table", None) is None and self.bank_nofract_safe():
                w.exciter_opts |= _lib.EXCITER_BANK_NOFRACT
            w.newt_out_w = P(m.newt.mixer[0].weight, "newt.mixer.0.weight", 64)
            w.newt_out_b = P(m.newt.mixer[0].bias, "newt.mixer.0.bias", 1)
            w.noise_window = P(m.noise_synth.window, "noise_synth.window", 256)
            # FIR design matrix (weights-independent apart from the window buffer) and, when every layer input provably
            # stays inside fp16 range, the pre-split fp16 fragment table that moves the frame MLPs to the fp16 matrix pipe
            fd = torch.empty(_lib.FIR_LEN * _lib.FIR_DESIGN_COLS, dtype=torch.float32, device=dev)
            check(L.nws_fir_design_matrix(w.noise_window, ptr(fd), st), "nws_fir_design_matrix")
            self._fir_design = fd
            keep.append(fd)
            w.mlp_frags = None
            if self.fp16_mlp_safe():
                frags = torch.empty(_lib.MLP_FRAGS_BYTES, dtype=torch.uint8, device=dev)
                check(L.nws_mlp_frags(C.byref(w), ptr(fd), ptr(frags), st), "nws_mlp_frags")
                keep.append(frags)
                w.mlp_frags = frags.data_ptr()
            keep.append(_req(m.osc.rand_phase.detach(), "osc.rand_phase", 101))
            keep.append(_req(m.reverb.ir.detach(), "reverb.ir"))
            devs = {t.device for t in keep}
            if len(devs) != 1:
                raise RuntimeError(f"model parameters are spread over several devices: {devs}")
            torch.cuda.current_stream(dev).synchronize()   # derived tables complete before any other stream can use them
        wdesc = torch.frombuffer(bytearray(bytes(w)), dtype=torch.uint8)      # the struct as the op layer takes it
        self._w = (w, keep, dev, wdesc)
        self._fp = fp
        return self._w

    HYBRID_W_BOUND = 1e-5     # worst-case output error (a tenth of the 1e-4 parity bar) under which "auto" may drop mixer terms

    def exciter_opts(self) -> int:
        """NwsWeights.exciter_opts.  `model.exciter_opts` if set, else the NWS_EXCITER_OPTS environment variable, else
        "auto" = 0 (every product of the 101 -> 64 mixer as a two-term fp16 split of both operands: 22-bit, fp32-class, what
        the reference's fp32 Conv1d at models/neural_waveshaping.py:54,66 is compared with) UNLESS the worst-case bound of
        `precision.hybrid_w_error_bound` - from the weights alone, valid for any input - proves that running harmonics 16..101
        as plain fp16 x fp16 products (EXCITER_HYBRID_W: 1 instead of 3 MFMAs per product there) cannot move the output by
        more than HYBRID_W_BOUND.  The bound multiplies worst-case FiLM gains, LUT slope and ||ir||_1: for the three shipped
        checkpoints it is 4e4 .. 9e4, so they run two-term; the hybrid forms remain an explicit opt-in (measured 3e-7 ..
        4e-6 RMS against the reference on the golden vectors, tests/test_gpu_parity.py)."""
        v = getattr(self._model_ref, "exciter_opts", None)
        if v is None:
            v = os.environ.get("NWS_EXCITER_OPTS")
        if v is not None and str(v) != "auto":
            return int(v)
        b = self.hybrid_w_bound()
        return _lib.EXCITER_HYBRID_W if (b is not None and b["bound"] <= self.HYBRID_W_BOUND) else 0

    def bank_nofract_safe(self, limit_turns: float = 128.0) -> bool:
        """Exact sin-MLP shapers: may the sines of the hidden and output layers go to v_sin_f32 without a v_fract in front?
        Their inputs are sines, so a row's pre-activation is bounded by sum |W| + |b| whatever the signal; v_sin_f32 reduces
        arguments inside +-256 turns itself.  One-time host check from the weights (the shipped checkpoints: 0.6 turns; the bound
        is held to half the domain).  NWS_BANK_FRACT=1 keeps the v_fract for A/B timing."""
        if os.environ.get("NWS_BANK_FRACT"):
            return False
        sh = self._model_ref.newt._modules.get("shaping_fn")
        if sh is None or getattr(sh, "depth", 0) != 4:
            return False
        with torch.no_grad():
            worst = 0.0
            for i in (2, 4, 6):
                wt, b = sh.net[i].weight.detach(), sh.net[i].bias.detach()
                worst = max(worst, float((wt.abs().flatten(1).sum(1) + b.abs()).max()) / (2.0 * math.pi))
        return math.isfinite(worst) and worst < limit_turns

    def hybrid_w_bound(self):
        """precision.hybrid_w_error_bound for this model (None for exact shapers: the option only exists on the LUT path)"""
        from . import precision

        m = self._model_ref
        table = getattr(m.newt, "lookup_table", None)
        if table is None:
            return None
        return precision.hybrid_w_error_bound(m, table, float(m.newt.table_min), float(m.newt.table_max))

    def fp16_mlp_safe(self, limit: float = 3.0e4) -> bool:
        """Worst-case magnitude of every frame-MLP layer input, from weight norms (one-time host check).
        |gru_out| < 1; emb rows <= ||W_proj||_1 + |b|; LayerNorm outputs <= sqrt(C-1)|gamma| + |beta|;
        H rows <= ||W_9||_1 * ln_bound + |b_9|.  All must stay well inside fp16 range for the two-term split."""
        m = self._model_ref
        with torch.no_grad():
            def l1(wt, b, scale):
                return float((wt.detach().abs().flatten(1).sum(1) * scale + b.detach().abs()).max())

            def ln(mod):
                return float(math.sqrt(_lib.HIDDEN - 1) * mod.layer_norm.weight.detach().abs().max()
                             + mod.layer_norm.bias.detach().abs().max())

            bounds = [l1(m.embedding.proj.weight, m.embedding.proj.bias, 1.0)]
            for mlp in (m.newt.mlp, m.h_generator):
                bounds += [ln(mlp.net[1]), ln(mlp.net[4]), ln(mlp.net[7])]
            bounds.append(l1(m.h_generator.net[9].weight, m.h_generator.net[9].bias, ln(m.h_generator.net[7])))
            wmax = max(float(p.detach().abs().max()) for mlp in (m.newt.mlp, m.h_generator) for p in mlp.parameters())
        return max(bounds + [wmax]) < limit and all(b == b for b in bounds)

    @property
    def device(self):
        return self.weights()[2]

    def rand_phase(self):
        return self.weights()[1][-2]

    def ir(self):
        return self.weights()[1][-1]

    # ---- weight-independent / cached tables ----------------------------------------------------
    def fir_design(self):
        self.weights()
        return self._fir_design

    def osc_sample_rate(self) -> float:
        """The rate the oscillator divides by: HarmonicOscillator's own gin binding (generators.py:41,59), which a configuration
        may set apart from NeuralWaveshaping.sample_rate - the reference's forward only ever uses the oscillator's."""
        m = self._model_ref
        return float(getattr(getattr(m, "osc", None), "sample_rate", m.sample_rate))

    def reverb_aux(self, n_samples: int):
        """(plan, tables, spectrum, plan tensor) for a forward of n_samples; the IR spectrum is rebuilt with the weights"""
        self._wd()
        return self._reverb_aux(n_samples)

    def _reverb_aux(self, n_samples: int):     # after _wd(): no second fingerprint walk
        ir = self._w[1][-1]
        dev = self._w[2]
        plan, tables, plan_t = reverb_plan_and_tables(dev, n_samples, ir.numel() + 1)
        skey = (plan.L, plan.N1, plan.N2)      # stored in the transform's own (k1, k2) order: one spectrum per factorisation
        spec = self._spectra.get(skey)
        if spec is None:
            with torch.cuda.device(dev):
                spec = torch.empty(_lib.lib().nws_reverb_spectrum_bytes(C.byref(plan)) // 4, dtype=torch.float32, device=dev)
                nbytes = _lib.lib().nws_reverb_workspace_bytes(C.byref(plan), 1)
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                check(_lib.lib().nws_reverb_ir_spectrum(C.byref(plan), ptr(tables), ptr(ir), ir.numel(), ptr(spec), ptr(ws),
                                                        nbytes, stream_ptr(dev)), "nws_reverb_ir_spectrum")
                torch.cuda.current_stream(dev).synchronize()
            self._spectra[skey] = spec
        return plan, tables, spec, plan_t

    # ---- stage launchers (public for the parity tests; each is one C-ABI call / one torch op) ------------------
    def phase_carry(self, f0=None, f0_up=None):
        src = f0 if f0 is not None else f0_up
        o = ops()
        if o is not None:
            return o.phase_carry(f0, f0_up)
        B = src.shape[0]
        T = f0.shape[-1] if f0 is not None else f0_up.shape[-1] // _lib.HOP
        with torch.cuda.device(src.device):
            carry = torch.empty((B, T * _lib.HOP // 32), dtype=torch.float64, device=src.device)
            check(_lib.lib().nws_phase_carry(ptr(f0), ptr(f0_up), B, T, ptr(carry), stream_ptr(src.device)), "nws_phase_carry")
        return carry

    def exciter_newt(self, f0, f0_up, carry, phase_u, film, want_exciter=False, want_newt=True):
        w, _, dev, wdesc = self._wd()
        src = f0 if f0 is not None else f0_up
        same_device(dev, f0=src, carry=carry, phase_u=phase_u, film=film)
        sr = self.osc_sample_rate()
        o = ops()
        if o is not None:
            exc, out = o.exciter_newt(wdesc, f0, f0_up, carry, phase_u, self._w[1][-2], film, sr, want_exciter, want_newt)
            return (exc if want_exciter else None), (out if want_newt else None)
        B = src.shape[0]
        T = f0.shape[-1] if f0 is not None else f0_up.shape[-1] // _lib.HOP
        N = T * _lib.HOP
        with torch.cuda.device(dev):
            exc = torch.empty((B, _lib.N_SHAPERS, N), dtype=torch.float32, device=dev) if want_exciter else None
            out = torch.empty((B, N), dtype=torch.float32, device=dev) if want_newt else None
            check(_lib.lib().nws_exciter_newt(C.byref(w), ptr(f0), ptr(f0_up), ptr(carry), ptr(phase_u), ptr(self._w[1][-2]),
                                              ptr(film), B, T, sr, ptr(exc), ptr(out), stream_ptr(dev)), "nws_exciter_newt")
        return exc, out

    def control_gru(self, control, batched=False, h0=None, return_state=False):
        w, _, dev, wdesc = self._wd()
        same_device(dev, control=control, h0=h0)
        B, Cc, T = control.shape
        o = ops()
        if o is not None:
            out, hT = o.control_gru(wdesc, control, h0, bool(batched))
            return (out, hT) if return_state else out
        with torch.cuda.device(dev):
            out = torch.empty((B, T, _lib.HIDDEN), dtype=torch.float32, device=dev)
            hT = torch.empty((B, _lib.HIDDEN), dtype=torch.float32, device=dev) if return_state else None
            fn = _lib.lib().nws_control_gru_batched if batched else _lib.lib().nws_control_gru_state
            check(fn(C.byref(w), ptr(control), B, Cc, T, ptr(h0), ptr(out), ptr(hT), stream_ptr(dev)), "nws_control_gru")
        return (out, hT) if return_state else out

    def frame_mlps(self, gru_out, want_emb=False, want_H=False):
        w, _, dev, wdesc = self._wd()
        same_device(dev, gru_out=gru_out)
        B, T, _ = gru_out.shape
        o = ops()
        if o is not None:
            emb, film, H, fir = o.frame_mlps(wdesc, gru_out, self._fir_design, want_emb, want_H)
            return (emb if want_emb else None), film, (H if want_H else None), fir
        with torch.cuda.device(dev):
            emb = torch.empty((B, _lib.HIDDEN, T), dtype=torch.float32, device=dev) if want_emb else None
            film = torch.empty((B, T, _lib.FILM_CH), dtype=torch.float32, device=dev)
            H = torch.empty((B, T, _lib.N_BANDS), dtype=torch.float32, device=dev) if want_H else None
            fir = torch.empty((B, T, _lib.FIR_HALF), dtype=torch.float32, device=dev)
            check(_lib.lib().nws_frame_mlps(C.byref(w), ptr(gru_out), ptr(self._fir_design), B, T, ptr(emb), ptr(film),
                                            ptr(H), ptr(fir), stream_ptr(dev)), "nws_frame_mlps")
        return emb, film, H, fir

    def fir_noise(self, fir, noise, add_in=None, origin=None, noise_len=None):
        """origin None: the reference's framing (N-1 noise samples, reflect padding); else a streaming window whose frame t
        covers noise[128 t - origin, +256) of the first `noise_len` samples (nws_fir_noise_window)"""
        B, T, _ = fir.shape
        same_device(fir.device, noise=noise, add_in=add_in)
        o = ops()
        if o is not None:
            nz = noise if noise_len is None or noise_len == noise.numel() else noise[:noise_len]
            return o.fir_noise(fir, nz, add_in, -1 if origin is None else int(origin))
        with torch.cuda.device(fir.device):
            out = torch.empty((B, T * _lib.HOP), dtype=torch.float32, device=fir.device)
            if origin is None:
                check(_lib.lib().nws_fir_noise(ptr(fir), ptr(noise), ptr(add_in), B, T, ptr(out), stream_ptr(fir.device)),
                      "nws_fir_noise")
            else:
                n_len = noise.numel() if noise_len is None else int(noise_len)
                check(_lib.lib().nws_fir_noise_window(ptr(fir), ptr(noise), n_len, int(origin), ptr(add_in), B, T, ptr(out),
                                                      stream_ptr(fir.device)), "nws_fir_noise_window")
        return out

    def reverb(self, x):
        B, N = x.shape
        _, _, dev, _ = self._wd()
        same_device(dev, x=x)
        plan, tables, spec, plan_t = self._reverb_aux(N)
        o = ops()
        if o is not None:
            return o.reverb(plan_t, tables, spec, x)
        with torch.cuda.device(x.device):
            nbytes = _lib.lib().nws_reverb_workspace_bytes(C.byref(plan), B)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            y = torch.empty_like(x)
            check(_lib.lib().nws_reverb(C.byref(plan), ptr(tables), ptr(spec), ptr(x), B, N, ptr(y), ptr(ws), nbytes,
                                        stream_ptr(x.device)), "nws_reverb")
        return y

    def reverb_linear_chunk(self, plan_aux, x, tail_in):
        """streaming: y = x + wet[:M] + tail_in[:M]; returns (y, tail_out)"""
        plan, tables, spec, plan_t = plan_aux
        same_device(tables.device, x=x, tail_in=tail_in)
        o = ops()
        if o is not None:
            return o.reverb_linear_chunk(plan_t, tables, spec, x, tail_in)
        B, M = x.shape
        with torch.cuda.device(x.device):
            nfl = (2 * ((B + 1) // 2) + B) * plan.L
            ws = torch.empty(nfl, dtype=torch.float32, device=x.device)
            y, tail_out = torch.empty_like(x), torch.empty_like(tail_in)
            check(_lib.lib().nws_reverb_linear_chunk(C.byref(plan), ptr(tables), ptr(spec), ptr(x), B, M, ptr(tail_in),
                                                     ptr(tail_out), tail_in.shape[1], ptr(y), ptr(ws), nfl * 4,
                                                     stream_ptr(x.device)), "nws_reverb_linear_chunk")
        return y, tail_out

    def shaper_table(self, size, tmin, tmax):
        w, _, dev, wdesc = self._wd()
        o = ops()
        if o is not None:
            return o.shaper_table(wdesc, self._w[1][-2], int(size), float(tmin), float(tmax))
        with torch.cuda.device(dev):
            t = torch.empty((_lib.N_SHAPERS, size), dtype=torch.float32, device=dev)
            check(_lib.lib().nws_shaper_table(C.byref(w), int(size), float(tmin), float(tmax), ptr(t), stream_ptr(dev)),
                  "nws_shaper_table")
        return t

    def shaper_apply(self, x):
        w, _, dev, wdesc = self._wd()
        same_device(dev, x=x)
        o = ops()
        if o is not None:
            return o.shaper_apply(wdesc, x)
        B, S, N = x.shape
        with torch.cuda.device(dev):
            y = torch.empty_like(x)
            check(_lib.lib().nws_shaper_apply(C.byref(w), ptr(x), B, N, ptr(y), stream_ptr(dev)), "nws_shaper_apply")
        return y

    def newt_apply(self, exciter, film):
        """NEWT.forward on a materialised exciter: exciter (B, 64, N), film (B, 256, T) channel-major -> (B, 1, N)"""
        w, _, dev, wdesc = self._wd()
        same_device(dev, exciter=exciter, film=film)
        o = ops()
        if o is not None:
            return o.newt_apply(wdesc, exciter, film)
        B, _, N = exciter.shape
        T = film.shape[2]
        if film.shape[1] != _lib.FILM_CH or N != T * _lib.HOP:
            raise RuntimeError(f"NEWT: exciter {tuple(exciter.shape)} and FiLM parameters {tuple(film.shape)} disagree")
        with torch.cuda.device(dev):
            out = torch.empty((B, 1, N), dtype=torch.float32, device=dev)
            check(_lib.lib().nws_newt_apply(C.byref(w), ptr(exciter), ptr(film), B, T, ptr(out), stream_ptr(dev)), "nws_newt_apply")
        return out

    # ---- the forward in two halves (throughput pipeline, pipeline.py) ------------------------------------------------
    def new_workspace(self, B, T):
        _, _, dev, _ = self._wd()
        plan, _, _, _ = self._reverb_aux(T * _lib.HOP)
        return torch.empty(_lib.lib().nws_forward_workspace_bytes(C.byref(plan), B, T), dtype=torch.uint8, device=dev)

    def forward_control(self, f0, control, ws, batched_gru=True):
        """phase carries + GRU into the head of `ws`, on the current stream"""
        w, _, dev, wdesc = self._wd()
        same_device(dev, f0=f0, control=control, workspace=ws)
        o = ops()
        if o is not None:
            o.forward_control(wdesc, f0, control, ws, bool(batched_gru))
            return
        B, Cc, T = control.shape
        with torch.cuda.device(dev):
            check(_lib.lib().nws_forward_control(C.byref(w), ptr(f0), ptr(control), B, Cc, T, 1 if batched_gru else 0, ptr(ws),
                                                 ws.numel(), stream_ptr(dev)), "nws_forward_control")

    def forward_audio(self, f0, B, T, phase_u, noise, ws, out=None, wait_event=None, record_event=None, row_blocks=None,
                      on_block=None, block_events=None):
        """frame MLPs .. reverb from the head of `ws` (forward_control must have completed in stream order / by event).
        wait_event / record_event: torch.cuda.Event hooks right before / after the oscillator kernel (nws_forward_audio_ev).
        row_blocks = [(row0, nrows), ...] (even row0, nrows >= 4): the reverb runs block by block and `on_block(row0, nrows,
        out)` is called after each block has been enqueued - a multi-GPU caller pushes that sub-batch to its peers while the
        next block's reverb runs.  Same bits as the single call."""
        w, _, dev, wdesc = self._wd()
        same_device(dev, f0=f0, phase_u=phase_u, noise=noise, workspace=ws, out=out)
        if block_events is not None and (row_blocks is None or len(row_blocks) < 2 or on_block is not None):
            # never silently ignored: synchronize() on a never-recorded event returns at once, and whoever waits on these events
            # (a helper thread that pushes sub-batches to peers) would read rows the reverb has not written yet
            raise RuntimeError("block_events are recorded by the one-call block path only: pass row_blocks of two or more blocks and "
                               "no on_block callback")
        N = T * _lib.HOP
        plan, tables, spec, plan_t = self._reverb_aux(N)
        sr = self.osc_sample_rate()
        o = ops()
        if row_blocks is not None and len(row_blocks) > 1:
            # the blocks must tile [0, B) with even sizes (two utterances share one transform; ForwardPipeline.row_blocks' rule):
            # anything else would leave rows of `out` unwritten or pair the wrong utterances
            nxt = 0
            for k, (row0, nrows) in enumerate(row_blocks):
                # every block but the last has an even size (so that every row0 is even: nws_forward_reverb_rows' own rule; an odd
                # last block pads its last pair like an odd batch does)
                if int(row0) != nxt or int(nrows) <= 0 or (int(nrows) % 2 and k != len(row_blocks) - 1):
                    raise RuntimeError(f"row_blocks must tile [0, {B}) in order, even sizes except the last, got {list(row_blocks)}")
                nxt += int(nrows)
            if nxt != B:
                raise RuntimeError(f"row_blocks cover {nxt} of {B} rows: {list(row_blocks)}")
            if wait_event is not None or record_event is not None:
                raise RuntimeError("wait_event / record_event hook the single-call form; they cannot be combined with row_blocks")
            if block_events is not None and on_block is None:
                # ONE call: audio_pre + every block's reverb, block q's torch.cuda.Event recorded behind it by the library itself
                # (nws_forward_audio_blocks; a helper thread waits on the events and pushes each sub-batch as it completes)
                if len(block_events) != len(row_blocks):
                    raise RuntimeError("block_events: one torch.cuda.Event per row block")
                with torch.cuda.device(dev):
                    if out is None:
                        out = torch.empty((B, N), dtype=torch.float32, device=dev)
                    for ev in block_events:         # (a never-recorded torch event has no handle yet: record() creates it lazily)
                        if not ev.cuda_event:
                            ev.record()
                    r0 = [int(r) for r, _ in row_blocks]
                    nr = [int(n) for _, n in row_blocks]
                    evs = [int(ev.cuda_event) for ev in block_events]
                    if o is not None:
                        o.forward_audio_blocks(wdesc, f0, phase_u, self._w[1][-2], noise, self._fir_design, plan_t, tables, spec, ws, sr, out,
                                               r0, nr, evs)
                    else:
                        aux = NwsForwardAux()
                        aux.fir_design = ptr(self._fir_design)
                        aux.plan = C.pointer(plan)
                        aux.reverb_tables = ptr(tables)
                        aux.reverb_spectrum = ptr(spec)
                        n = len(r0)
                        check(_lib.lib().nws_forward_audio_blocks(C.byref(w), C.byref(aux), ptr(f0), B, T, sr, ptr(phase_u), ptr(self._w[1][-2]),
                                                                  ptr(noise), ptr(out), ptr(ws), ws.numel(), stream_ptr(dev),
                                                                  (C.c_int32 * n)(*r0), (C.c_int32 * n)(*nr), (C.c_void_p * n)(*evs), n),
                              "nws_forward_audio_blocks")
                return out
            with torch.cuda.device(dev):
                if out is None:
                    out = torch.empty((B, N), dtype=torch.float32, device=dev)
                aux = None
                if o is not None:
                    o.forward_audio_pre(wdesc, f0, phase_u, self._w[1][-2], noise, self._fir_design, plan_t, tables, spec, ws, sr)
                else:
                    aux = NwsForwardAux()
                    aux.fir_design = ptr(self._fir_design)
                    aux.plan = C.pointer(plan)
                    aux.reverb_tables = ptr(tables)
                    aux.reverb_spectrum = ptr(spec)
                    check(_lib.lib().nws_forward_audio_pre(C.byref(w), C.byref(aux), ptr(f0), B, T, sr, ptr(phase_u), ptr(self._w[1][-2]),
                                                           ptr(noise), ptr(ws), ws.numel(), stream_ptr(dev)), "nws_forward_audio_pre")
                for row0, nrows in row_blocks:
                    if o is not None:
                        o.forward_reverb_rows(self._fir_design, plan_t, tables, spec, ws, T, int(row0), int(nrows), out)
                    else:
                        check(_lib.lib().nws_forward_reverb_rows(C.byref(aux), B, T, int(row0), int(nrows), ptr(out), ptr(ws), ws.numel(),
                                                                 stream_ptr(dev)), "nws_forward_reverb_rows")
                    if on_block is not None:
                        on_block(int(row0), int(nrows), out)
            return out
        if o is not None:
            return o.forward_audio(wdesc, f0, phase_u, self._w[1][-2], noise, self._fir_design, plan_t, tables, spec, ws, sr, out,
                                   wait_event.cuda_event if wait_event is not None else 0,
                                   record_event.cuda_event if record_event is not None else 0)
        with torch.cuda.device(dev):
            if out is None:
                out = torch.empty((B, N), dtype=torch.float32, device=dev)
            aux = NwsForwardAux()
            aux.fir_design = ptr(self._fir_design)
            aux.plan = C.pointer(plan)
            aux.reverb_tables = ptr(tables)
            aux.reverb_spectrum = ptr(spec)
            check(_lib.lib().nws_forward_audio_ev(C.byref(w), C.byref(aux), ptr(f0), B, T, sr, ptr(phase_u), ptr(self._w[1][-2]),
                                                  ptr(noise), ptr(out), ptr(ws), ws.numel(), stream_ptr(dev),
                                                  wait_event.cuda_event if wait_event is not None else None,
                                                  record_event.cuda_event if record_event is not None else None),
                  "nws_forward_audio_ev")
        return out

    # ---- the whole forward: ONE op / ONE C-ABI call --------------------------------------------------------
    def forward(self, f0, control, phase_u, noise):
        if not self.specialised():
            return self.generic.forward(f0, control, phase_u, noise)
        w, _, dev, wdesc = self._wd()
        same_device(dev, f0=f0, control=control, phase_u=phase_u, noise=noise)
        B, Cc, T = control.shape
        N = T * _lib.HOP
        plan, tables, spec, plan_t = self._reverb_aux(N)
        key = (B, T, stream_ptr(dev))   # one scratch arena per stream: forwards on different streams may overlap
        ws = self._workspaces.get(key)
        if ws is None:
            nbytes = _lib.lib().nws_forward_workspace_bytes(C.byref(plan), B, T)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            if len(self._workspaces) >= 16:
                # forget the oldest entry only: arenas of other streams may still be in use by enqueued work (their memory
                # is handed back to the caching allocator, which re-issues a block only in the stream order it was used in)
                self._workspaces.pop(next(iter(self._workspaces)))
            self._workspaces[key] = ws
        sr = self.osc_sample_rate()
        o = ops()
        if o is not None:
            return o.forward(wdesc, f0, control, phase_u, self._w[1][-2], noise, self._fir_design, plan_t, tables, spec, ws, sr)
        with torch.cuda.device(dev):
            out = torch.empty((B, N), dtype=torch.float32, device=dev)
            aux = NwsForwardAux()
            aux.fir_design = ptr(self._fir_design)
            aux.plan = C.pointer(plan)
            aux.reverb_tables = ptr(tables)
            aux.reverb_spectrum = ptr(spec)
            check(_lib.lib().nws_forward(C.byref(w), C.byref(aux), ptr(f0), ptr(control), B, Cc, T, sr, ptr(phase_u),
                                         ptr(self._w[1][-2]), ptr(noise), ptr(out), ptr(ws), ws.numel(), stream_ptr(dev)),
                  "nws_forward")
        return out
