"""Host-side engine: turns a NeuralWaveshaping module's parameters into the C-ABI's pointer struct,
owns the weight-independent tables (FIR design matrix, reverb DFT tables) and the cached IR spectrum,
and enqueues the HIP kernels on torch's current stream.

PyTorch is used here only as plumbing: device memory (tensors), the current HIP stream and the
device RNG that the reference itself draws from inside forward().
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from ._lib import NwsForwardAux, NwsReverbPlan, NwsWeights, check, ptr, stream_ptr


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


_TABLE_CACHE: dict = {}  # (device index, L) -> (plan, tables tensor)


def reverb_plan_and_tables(device: torch.device, n_samples: int, ir_len_plus1: int):
    plan = NwsReverbPlan()
    check(_lib.lib().nws_reverb_plan(int(n_samples), int(ir_len_plus1), C.byref(plan)), "nws_reverb_plan")
    key = (device.index, plan.L)
    hit = _TABLE_CACHE.get(key)
    if hit is None:
        nbytes = _lib.lib().nws_reverb_table_bytes(C.byref(plan))
        tables = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        check(_lib.lib().nws_reverb_build_tables(C.byref(plan), ptr(tables), stream_ptr()), "nws_reverb_build_tables")
        # one-time: make the tables visible to every stream before anybody can use them (callers may issue forwards
        # round-robin on several streams; the builder stream is whichever one got here first)
        torch.cuda.current_stream().synchronize()
        hit = (plan, tables)
        _TABLE_CACHE[key] = hit
    return hit


class Engine:
    """Per-model launcher.  ``model`` is a NeuralWaveshaping module (see models/neural_waveshaping.py)."""

    def __init__(self, model):
        self._model_ref = model
        self._w = None          # (NwsWeights, keep-alive list, device)
        self._fir_design = None
        self._spectra = {}      # L -> spectrum tensor
        self._workspaces = {}   # (B, T) -> tensor

    # ---- cache control -----------------------------------------------------------------------
    def invalidate(self):
        self._w = None
        self._fir_design = None
        self._spectra.clear()
        self._workspaces.clear()

    # ---- weights -----------------------------------------------------------------------------
    def weights(self):
        if self._w is not None:
            return self._w
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
        frags = torch.empty(28672, dtype=torch.uint8, device=keep[-1].device)
        check(_lib.lib().nws_mixer_frags(w.mixer_w, w.mixer_b, ptr(frags), stream_ptr()), "nws_mixer_frags")
        keep.append(frags)
        w.mixer_frags = frags.data_ptr()
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
            turns = torch.empty((64, _lib.SHAPER_TURNS_ROW), dtype=torch.float32, device=keep[-1].device)
            check(_lib.lib().nws_shaper_turns(C.byref(w), ptr(turns), stream_ptr()), "nws_shaper_turns")
            keep.append(turns)
            w.shaper_turns = turns.data_ptr()
        table = getattr(m.newt, "lookup_table", None)
        if table is not None:
            size = int(m.newt.table_size)
            w.lut = P(table, "newt.lookup_table", 64 * size)
            pairs = torch.empty((64, size, 2), dtype=torch.float32, device=table.device)
            check(_lib.lib().nws_lut_pairs(w.lut, size, ptr(pairs), stream_ptr()), "nws_lut_pairs")
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
        w.newt_out_w = P(m.newt.mixer[0].weight, "newt.mixer.0.weight", 64)
        w.newt_out_b = P(m.newt.mixer[0].bias, "newt.mixer.0.bias", 1)
        w.noise_window = P(m.noise_synth.window, "noise_synth.window", 256)
        # FIR design matrix (weights-independent apart from the window buffer) and, when every layer input provably
        # stays inside fp16 range, the pre-split fp16 fragment table that moves the frame MLPs to the fp16 matrix pipe
        fd = torch.empty(_lib.FIR_LEN * _lib.FIR_DESIGN_COLS, dtype=torch.float32, device=keep[0].device)
        check(_lib.lib().nws_fir_design_matrix(w.noise_window, ptr(fd), stream_ptr()), "nws_fir_design_matrix")
        self._fir_design = fd
        keep.append(fd)
        w.mlp_frags = None
        if self.fp16_mlp_safe():
            frags = torch.empty(819200, dtype=torch.uint8, device=fd.device)
            check(_lib.lib().nws_mlp_frags(C.byref(w), ptr(fd), ptr(frags), stream_ptr()), "nws_mlp_frags")
            keep.append(frags)
            w.mlp_frags = frags.data_ptr()
        keep.append(_req(m.osc.rand_phase.detach(), "osc.rand_phase", 101))
        keep.append(_req(m.reverb.ir.detach(), "reverb.ir"))
        devs = {t.device for t in keep}
        if len(devs) != 1:
            raise RuntimeError(f"model parameters are spread over several devices: {devs}")
        torch.cuda.current_stream().synchronize()   # derived tables complete before any other stream can use them
        self._w = (w, keep, next(iter(devs)))
        return self._w

    def exciter_opts(self) -> int:
        """NwsWeights.exciter_opts.  `model.exciter_opts` if set (call invalidate_cache() after changing it), else the
        NWS_EXCITER_OPTS environment variable, else "auto": the sines of harmonics 16..101 travel as ONE fp16 term
        (EXCITER_HYBRID: 2 instead of 3 MFMAs per product there, no residual split) when the mixer bias + harmonics 1..15 hold
        at least 55 % of the harmonic mixer's weight energy - true for the three shipped checkpoints (61 / 67 / 85 %), where
        it costs 2-6e-6 RMS end to end (tests/test_gpu_parity.py holds it to 1e-5 on every golden vector; the bar is 1e-4);
        otherwise (e.g. random initialisation) every sine keeps two terms."""
        import os

        v = getattr(self._model_ref, "exciter_opts", None)
        if v is None:
            v = os.environ.get("NWS_EXCITER_OPTS")
        if v is not None and str(v) != "auto":
            return int(v)
        with torch.no_grad():
            e = self._model_ref.harmonic_mixer.weight.detach().float().pow(2).sum(dim=(0, 2))     # per harmonic
            share = float(e[:15].sum() / e.sum().clamp_min(1e-30))
        return _lib.EXCITER_HYBRID if share >= 0.55 else 0

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

    def reverb_aux(self, n_samples: int):
        ir = self.ir()
        plan, tables = reverb_plan_and_tables(self.device, n_samples, ir.numel() + 1)
        spec = self._spectra.get(plan.L)
        if spec is None:
            spec = torch.empty(_lib.lib().nws_reverb_spectrum_bytes(C.byref(plan)) // 4, dtype=torch.float32,
                               device=self.device)
            nbytes = _lib.lib().nws_reverb_workspace_bytes(C.byref(plan), 1)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            check(_lib.lib().nws_reverb_ir_spectrum(C.byref(plan), ptr(tables), ptr(ir), ir.numel(), ptr(spec), ptr(ws),
                                                    nbytes, stream_ptr()), "nws_reverb_ir_spectrum")
            torch.cuda.current_stream().synchronize()
            self._spectra[plan.L] = spec
        return plan, tables, spec

    # ---- stage launchers (public for the parity tests; each is one C-ABI call) ------------------
    def phase_carry(self, f0=None, f0_up=None):
        src = f0 if f0 is not None else f0_up
        B = src.shape[0]
        T = f0.shape[-1] if f0 is not None else f0_up.shape[-1] // _lib.HOP
        carry = torch.empty((B, T * _lib.HOP // 32), dtype=torch.float64, device=src.device)
        check(_lib.lib().nws_phase_carry(ptr(f0), ptr(f0_up), B, T, ptr(carry), stream_ptr()), "nws_phase_carry")
        return carry

    def exciter_newt(self, f0, f0_up, carry, phase_u, film, want_exciter=False, want_newt=True):
        w, _, dev = self.weights()
        src = f0 if f0 is not None else f0_up
        B = src.shape[0]
        T = f0.shape[-1] if f0 is not None else f0_up.shape[-1] // _lib.HOP
        N = T * _lib.HOP
        exc = torch.empty((B, _lib.N_SHAPERS, N), dtype=torch.float32, device=dev) if want_exciter else None
        out = torch.empty((B, N), dtype=torch.float32, device=dev) if want_newt else None
        m = self._model_ref
        check(_lib.lib().nws_exciter_newt(C.byref(w), ptr(f0), ptr(f0_up), ptr(carry), ptr(phase_u),
                                          ptr(self.rand_phase()), ptr(film), B, T, float(m.sample_rate), ptr(exc),
                                          ptr(out), stream_ptr()), "nws_exciter_newt")
        return exc, out

    def control_gru(self, control, batched=False):
        w, _, dev = self.weights()
        B, Cc, T = control.shape
        out = torch.empty((B, T, _lib.HIDDEN), dtype=torch.float32, device=dev)
        if batched:
            check(_lib.lib().nws_control_gru_batched(C.byref(w), ptr(control), B, Cc, T, None, ptr(out), None,
                                                     stream_ptr()), "nws_control_gru_batched")
        else:
            check(_lib.lib().nws_control_gru(C.byref(w), ptr(control), B, Cc, T, ptr(out), stream_ptr()), "nws_control_gru")
        return out

    def frame_mlps(self, gru_out, want_emb=False, want_H=False):
        w, _, dev = self.weights()
        B, T, _ = gru_out.shape
        emb = torch.empty((B, _lib.HIDDEN, T), dtype=torch.float32, device=dev) if want_emb else None
        film = torch.empty((B, T, _lib.FILM_CH), dtype=torch.float32, device=dev)
        H = torch.empty((B, T, _lib.N_BANDS), dtype=torch.float32, device=dev) if want_H else None
        fir = torch.empty((B, T, _lib.FIR_LEN), dtype=torch.float32, device=dev)
        check(_lib.lib().nws_frame_mlps(C.byref(w), ptr(gru_out), ptr(self.fir_design()), B, T, ptr(emb), ptr(film),
                                        ptr(H), ptr(fir), stream_ptr()), "nws_frame_mlps")
        return emb, film, H, fir

    def fir_noise(self, fir, noise, add_in=None):
        B, T, _ = fir.shape
        out = torch.empty((B, T * _lib.HOP), dtype=torch.float32, device=fir.device)
        check(_lib.lib().nws_fir_noise(ptr(fir), ptr(noise), ptr(add_in), B, T, ptr(out), stream_ptr()), "nws_fir_noise")
        return out

    def reverb(self, x):
        B, N = x.shape
        plan, tables, spec = self.reverb_aux(N)
        nbytes = _lib.lib().nws_reverb_workspace_bytes(C.byref(plan), B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        y = torch.empty_like(x)
        check(_lib.lib().nws_reverb(C.byref(plan), ptr(tables), ptr(spec), ptr(x), B, N, ptr(y), ptr(ws), nbytes,
                                    stream_ptr()), "nws_reverb")
        return y

    def shaper_table(self, size, tmin, tmax):
        w, _, dev = self.weights()
        t = torch.empty((_lib.N_SHAPERS, size), dtype=torch.float32, device=dev)
        check(_lib.lib().nws_shaper_table(C.byref(w), int(size), float(tmin), float(tmax), ptr(t), stream_ptr()),
              "nws_shaper_table")
        return t

    def shaper_apply(self, x):
        w, _, _ = self.weights()
        B, S, N = x.shape
        y = torch.empty_like(x)
        check(_lib.lib().nws_shaper_apply(C.byref(w), ptr(x), B, N, ptr(y), stream_ptr()), "nws_shaper_apply")
        return y

    # ---- the whole forward: ONE C-ABI call --------------------------------------------------------
    # ---- the forward in two halves (throughput pipeline, pipeline.py) ------------------------------------------------
    def new_workspace(self, B, T):
        plan, _, _ = self.reverb_aux(T * _lib.HOP)
        _, _, dev = self.weights()
        return torch.empty(_lib.lib().nws_forward_workspace_bytes(C.byref(plan), B, T), dtype=torch.uint8, device=dev)

    def forward_control(self, f0, control, ws, batched_gru=True):
        """phase carries + GRU into the head of `ws`, on the current stream"""
        w, _, _ = self.weights()
        B, Cc, T = control.shape
        check(_lib.lib().nws_forward_control(C.byref(w), ptr(f0), ptr(control), B, Cc, T, 1 if batched_gru else 0, ptr(ws),
                                             ws.numel(), stream_ptr()), "nws_forward_control")

    def forward_audio(self, f0, B, T, phase_u, noise, ws, out=None):
        """frame MLPs .. reverb from the head of `ws` (forward_control must have completed in stream order / by event)"""
        w, _, dev = self.weights()
        N = T * _lib.HOP
        plan, tables, spec = self.reverb_aux(N)
        if out is None:
            out = torch.empty((B, N), dtype=torch.float32, device=dev)
        aux = NwsForwardAux()
        aux.fir_design = ptr(self.fir_design())
        aux.plan = C.pointer(plan)
        aux.reverb_tables = ptr(tables)
        aux.reverb_spectrum = ptr(spec)
        check(_lib.lib().nws_forward_audio(C.byref(w), C.byref(aux), ptr(f0), B, T, float(self._model_ref.sample_rate),
                                           ptr(phase_u), ptr(self.rand_phase()), ptr(noise), ptr(out), ptr(ws), ws.numel(),
                                           stream_ptr()), "nws_forward_audio")
        return out

    def forward(self, f0, control, phase_u, noise, out=None):
        w, _, dev = self.weights()
        B, Cc, T = control.shape
        N = T * _lib.HOP
        plan, tables, spec = self.reverb_aux(N)
        key = (B, T, stream_ptr())   # one scratch arena per stream: forwards on different streams may overlap
        ws = self._workspaces.get(key)
        if ws is None:
            nbytes = _lib.lib().nws_forward_workspace_bytes(C.byref(plan), B, T)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            if len(self._workspaces) > 8:
                self._workspaces.clear()
            self._workspaces[key] = ws
        if out is None:
            out = torch.empty((B, N), dtype=torch.float32, device=dev)
        aux = NwsForwardAux()
        aux.fir_design = ptr(self.fir_design())
        aux.plan = C.pointer(plan)
        aux.reverb_tables = ptr(tables)
        aux.reverb_spectrum = ptr(spec)
        check(_lib.lib().nws_forward(C.byref(w), C.byref(aux), ptr(f0), ptr(control), B, Cc, T,
                                     float(self._model_ref.sample_rate), ptr(phase_u), ptr(self.rand_phase()),
                                     ptr(noise), ptr(out), ptr(ws), ws.numel(), stream_ptr()), "nws_forward")
        return out
