"""Runtime-size path: host side of csrc/generic.hip.

The fused kernels are compiled for the one architecture the reference ships (gin/models/newt.gin).  The reference itself
builds any size its gin bindings name (models/neural_waveshaping.py:31-62, shaping.py:41-65, generators.py:11-48); those
configurations run here: the same C-ABI conventions, one plain-fp32 HIP stage kernel per launch (`nws_g_*`), chained by ONE
call (`nws_forward_generic` / `torch.ops.newt_hip.forward_generic`).  No PyTorch arithmetic: torch allocates the workspace
and provides the stream, nothing else.
"""
from __future__ import annotations

import ctypes as C
import itertools

import torch
import torch.nn as nn

from . import _lib
from ._lib import NwsGenericModel, NwsReverbPlan, NwsShaperDesc, check, ptr


def _p(t: torch.Tensor, name: str, numel: int | None = None) -> torch.Tensor:
    from .engine import _req

    t = _req(t.detach(), name)
    if numel is not None and t.numel() != numel:
        raise RuntimeError(f"{name}: expected {numel} elements, got {tuple(t.shape)}")
    return t


def shaper_desc(sh=None, *, lut=None, lut_min=0.0, lut_max=0.0, keep=None) -> NwsShaperDesc:
    """NwsShaperDesc of a TrainableNonlinearity (`sh`) or of a FastNEWT table (`lut` (S, size))."""
    d = NwsShaperDesc()
    keep = keep if keep is not None else []
    if lut is not None:
        t = _p(lut, "lookup_table")
        keep.append(t)
        d.n_shapers, d.lut_size = int(t.shape[0]), int(t.shape[1])
        d.lut_min, d.lut_max = float(lut_min), float(lut_max)
        d.lut = t.data_ptr()
        d.width, d.depth = 1, 1
        return d
    convs = [m for m in sh.net if isinstance(m, nn.Conv1d)]
    acts = [m for m in sh.net if not isinstance(m, nn.Conv1d)]
    if not all(type(a).__name__ == "Sine" for a in acts):
        raise RuntimeError("the HIP shapers implement the sine activations NEWT configures (nonlinearity=Sine)")
    depth, S, W = len(convs), int(sh.channels), int(sh.width)
    if depth < 1 or depth > 8:
        raise RuntimeError(f"TrainableNonlinearity depth {depth}: the HIP path takes 1..8 layers")
    d.n_shapers, d.width, d.depth = S, W, depth
    t = _p(sh.input_scale, "shaping_fn.input_scale", S)
    keep.append(t)
    d.in_scale = t.data_ptr()
    for i, c in enumerate(convs):
        rows = S if i == depth - 1 else S * W
        cols = 1 if i == 0 else W
        wt, bt = _p(c.weight, f"shaping_fn.net.{2 * i}.weight", rows * cols), _p(c.bias, f"shaping_fn.net.{2 * i}.bias", rows)
        keep += [wt, bt]
        d.w[i], d.b[i] = wt.data_ptr(), bt.data_ptr()
    return d


def desc_bytes(struct) -> torch.Tensor:
    """a ctypes struct as the CPU uint8 tensor the torch.ops layer takes"""
    return torch.frombuffer(bytearray(bytes(struct)), dtype=torch.uint8)


class GenericEngine:
    """Launcher of the runtime-size forward for one NeuralWaveshaping module (any gin configuration)."""

    def __init__(self, model):
        self._model_ref = model
        self._cache = None      # (fingerprint, NwsGenericModel, keep-alive list, device, gdesc tensor)
        self._reverb = {}       # N -> (plan | None, tables, spectrum, plan tensor)
        self._workspaces = {}

    def invalidate(self):
        self._cache = None
        self._reverb.clear()
        self._workspaces.clear()

    def _fingerprint(self):
        m = self._model_ref
        return tuple((t.data_ptr(), t._version) for t in itertools.chain(m.parameters(), m.buffers())) + (id(m.newt),)

    def model_desc(self):
        fp = self._fingerprint()
        if self._cache is not None and self._cache[0] == fp:
            return self._cache[1:]
        if self._cache is not None:
            torch.cuda.synchronize(self._cache[3])
            self.invalidate()
        m = self._model_ref
        keep = []

        def P(t, name, numel=None):
            t = _p(t, name, numel)
            keep.append(t)
            return t.data_ptr()

        g = NwsGenericModel()
        gru = m.embedding.gru
        if gru.num_layers != 1 or gru.bidirectional or not gru.batch_first or not gru.bias:
            raise RuntimeError("ControlModule: the HIP path implements nn.GRU(control_size, hidden_size, batch_first=True)")
        H, Cin = int(gru.hidden_size), int(gru.input_size)
        if Cin != 2:
            raise RuntimeError(f"ControlModule.control_size = {Cin}: NeuralWaveshaping.get_embedding always feeds control[:, 0:2] "
                               "(models/neural_waveshaping.py:69-72), so the reference's own forward only works with 2")
        E = int(m.embedding.proj.out_channels)
        S, K = int(m.harmonic_mixer.out_channels), int(m.osc.n_harmonics)
        hop = int(m.control_hop)
        g.control_size, g.gru_hidden, g.embedding, g.n_harmonics, g.n_shapers, g.hop = Cin, H, E, K, S, hop
        g.gru_w_ih, g.gru_w_hh = P(gru.weight_ih_l0, "gru.weight_ih_l0", 3 * H * Cin), P(gru.weight_hh_l0, "gru.weight_hh_l0", 3 * H * H)
        g.gru_b_ih, g.gru_b_hh = P(gru.bias_ih_l0, "gru.bias_ih_l0", 3 * H), P(gru.bias_hh_l0, "gru.bias_hh_l0", 3 * H)
        g.proj_w, g.proj_b = P(m.embedding.proj.weight, "embedding.proj.weight", E * H), P(m.embedding.proj.bias, "proj.bias", E)
        if m.harmonic_mixer.in_channels != K or m.newt.n_waveshapers != S:
            raise RuntimeError("harmonic_mixer / NEWT.n_waveshapers / HarmonicOscillator.n_harmonics disagree")
        g.mixer_w, g.mixer_b = P(m.harmonic_mixer.weight, "harmonic_mixer.weight", S * K), P(m.harmonic_mixer.bias, "mixer.bias", S)
        eps, slope = None, None
        for name, mlp, in_size, out_rows, wf, bf, gf, lf, depth_field, hid_field in (
                ("newt.mlp", m.newt.mlp, E, 4 * S, g.newt_mlp_w, g.newt_mlp_b, g.newt_ln_g, g.newt_ln_b, "newt_mlp_depth", None),
                ("h_generator", m.h_generator, E, None, g.hgen_w, g.hgen_b, g.hgen_ln_g, g.hgen_ln_b, "hgen_depth", "hgen_hidden")):
            convs = [c for c in mlp.net if isinstance(c, nn.Conv1d)]
            norms = [n for n in mlp.net if hasattr(n, "layer_norm")]
            acts = [a for a in mlp.net if isinstance(a, nn.LeakyReLU)]
            depth = len(convs)
            if depth < 1 or depth > 8 or len(norms) != depth - 1 or len(acts) != depth - 1:
                raise RuntimeError(f"{name}: unexpected layer stack")
            hidden = int(convs[0].out_channels) if depth > 1 else in_size
            if convs[0].in_channels != in_size:
                raise RuntimeError(f"{name}: in_size {convs[0].in_channels} != embedding_size {in_size}")
            if name == "newt.mlp" and (hidden != E or convs[-1].out_channels != 4 * S):
                raise RuntimeError("newt.mlp: expected TimeDistributedMLP(E, E, 4 * n_waveshapers) (shaping.py:53-55)")
            setattr(g, depth_field, depth)
            if hid_field:
                setattr(g, hid_field, hidden)
            for i, c in enumerate(convs):
                wf[i] = P(c.weight, f"{name}.net.{3 * i}.weight", c.out_channels * c.in_channels)
                bf[i] = P(c.bias, f"{name}.net.{3 * i}.bias", c.out_channels)
            for i, n in enumerate(norms):
                gf[i] = P(n.layer_norm.weight, f"{name} layer_norm.weight", hidden)
                lf[i] = P(n.layer_norm.bias, f"{name} layer_norm.bias", hidden)
                e_i, s_i = float(n.layer_norm.eps), float(acts[i].negative_slope)
                if (eps is not None and e_i != eps) or (slope is not None and s_i != slope):
                    raise RuntimeError("the frame MLPs must share one LayerNorm eps and one LeakyReLU slope")
                eps, slope = e_i, s_i
        g.ln_eps, g.leaky_slope = (eps if eps is not None else 1e-5), (slope if slope is not None else 0.01)
        ns = m.noise_synth
        L = int(ns.ir_length)
        if int(ns.hop_length) != hop:
            raise RuntimeError(f"FIRNoiseSynth.hop_length {ns.hop_length} != control_hop {hop}: the reference's torch.cat of the two "
                               "branches (models/neural_waveshaping.py:85) needs equal lengths")
        if L % 2 or L < hop:
            raise RuntimeError(f"FIRNoiseSynth.ir_length {L}: must be even and >= hop_length (torch.istft's overlap-add, generators.py:34)")
        hconvs = [c for c in m.h_generator.net if isinstance(c, nn.Conv1d)]
        if hconvs[-1].out_channels != L // 2 + 1:
            raise RuntimeError(f"h_generator.out_size {hconvs[-1].out_channels} != ir_length / 2 + 1 = {L // 2 + 1}")
        g.fir_len = L
        g.noise_window = P(ns.window, "noise_synth.window", L)
        mix = m.newt.mixer[0]
        g.out_channels = int(mix.out_channels)
        g.newt_out_w, g.newt_out_b = P(mix.weight, "newt.mixer.0.weight", g.out_channels * S), P(mix.bias, "newt.mixer.0.bias", g.out_channels)
        ir = _p(m.reverb.ir, "reverb.ir")
        g.ir, g.ir_len = ir.data_ptr(), int(ir.numel())
        table = getattr(m.newt, "lookup_table", None)
        if table is not None:
            g.shaper = shaper_desc(lut=table, lut_min=m.newt.table_min, lut_max=m.newt.table_max, keep=keep)
        else:
            g.shaper = shaper_desc(m.newt._modules["shaping_fn"], keep=keep)
        if g.shaper.n_shapers != S:
            raise RuntimeError("newt.shaping_fn / lookup_table and n_waveshapers disagree")
        rp = _p(m.osc.rand_phase.reshape(-1), "osc.rand_phase", K)
        keep += [rp, ir]
        devs = {t.device for t in keep}
        if len(devs) != 1:
            raise RuntimeError(f"model parameters are spread over several devices: {devs}")
        self._cache = (fp, g, keep, keep[0].device, desc_bytes(g))
        return self._cache[1:]

    def _reverb_aux(self, N, dev, ir):
        hit = self._reverb.get(N)
        if hit is None:
            from .engine import reverb_plan_and_tables

            plan = NwsReverbPlan()
            rc = _lib.lib().nws_reverb_plan(int(N), int(ir.numel()) + 1, C.byref(plan))
            if rc != 0:
                hit = (None, None, None, None)      # no factorisation: nws_g_reverb_direct
            else:
                plan, tables, plan_t = reverb_plan_and_tables(dev, N, ir.numel() + 1)
                L = _lib.lib()
                with torch.cuda.device(dev):
                    spec = torch.empty(L.nws_reverb_spectrum_bytes(C.byref(plan)) // 4, dtype=torch.float32, device=dev)
                    nb = L.nws_reverb_workspace_bytes(C.byref(plan), 1)
                    ws1 = torch.empty(nb, dtype=torch.uint8, device=dev)
                    check(L.nws_reverb_ir_spectrum(C.byref(plan), ptr(tables), ptr(ir), ir.numel(), ptr(spec), ptr(ws1), nb,
                                                   torch.cuda.current_stream(dev).cuda_stream), "nws_reverb_ir_spectrum")
                    torch.cuda.current_stream(dev).synchronize()
                hit = (plan, tables, spec, plan_t)
            self._reverb[N] = hit
        return hit

    def forward(self, f0, control, phase_u, noise):
        from .engine import ops, same_device, stream_ptr

        g, keep, dev, gdesc = self.model_desc()
        same_device(dev, f0=f0, control=control, phase_u=phase_u, noise=noise)
        B, Cc, T = control.shape
        N = T * g.hop
        ir, rp = keep[-1], keep[-2]
        plan, tables, spec, plan_t = self._reverb_aux(N, dev, ir)
        L = _lib.lib()
        key = (B, T, stream_ptr(dev))
        ws = self._workspaces.get(key)
        if ws is None:
            nbytes = L.nws_forward_generic_workspace_bytes(C.byref(g), B, T)
            rv_bytes = L.nws_reverb_workspace_bytes(C.byref(plan), B) if plan is not None else 0
            ws = (torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.empty(max(rv_bytes, 256), dtype=torch.uint8, device=dev))
            if len(self._workspaces) >= 8:
                self._workspaces.pop(next(iter(self._workspaces)))
            self._workspaces[key] = ws
        sr = float(getattr(getattr(self._model_ref, 'osc', None), 'sample_rate', self._model_ref.sample_rate))   # the oscillator's own binding (generators.py:41)
        o = ops()
        if o is not None:
            return o.forward_generic(gdesc, f0, control, phase_u, rp, noise, plan_t, tables, spec, ws[1], ws[0], sr)
        with torch.cuda.device(dev):
            out = torch.empty((B, N), dtype=torch.float32, device=dev)
            check(L.nws_forward_generic(C.byref(g), ptr(f0), ptr(control), B, Cc, T, sr, ptr(phase_u), ptr(rp), ptr(noise),
                                        C.byref(plan) if plan is not None else None, ptr(tables), ptr(spec),
                                        ptr(ws[1]) if plan is not None else None, ws[1].numel() if plan is not None else 0,
                                        ptr(out), ptr(ws[0]), ws[0].numel(), stream_ptr(dev)), "nws_forward_generic")
        return out
