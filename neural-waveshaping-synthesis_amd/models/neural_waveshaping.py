"""``NeuralWaveshaping`` - the drop-in module surface of the MI355X NEWT engine.

Mirrors the reference's ``NeuralWaveshaping`` (models/neural_waveshaping.py:29-90): same constructor
(gin-configurable), sub-module / attribute names, state-dict keys, ``forward(f0, control)``,
``render_exciter`` and ``get_embedding``.  ``forward`` is ONE call into the C-ABI (``nws_forward``),
which enqueues the hand-written HIP kernels on torch's current stream.  Training hooks of the
reference (:92-165) are out of scope.

Hidden inputs, exactly like the reference: every forward draws ``rand_like(osc.rand_phase)`` and then
``rand(control_hop*T - 1)`` from the default generator of the module's device, in that order
(generators.py:55, :30).  For parity testing the two draws can be injected with the keyword-only
arguments ``phase_u`` and ``noise``.
"""
import os

import torch
import torch.nn as nn

from .. import _lib
from .. import ginlite as gin
from ..engine import Engine, _req
from .modules import _standalone as sa
from .modules.dynamic import Conv1x1, TimeDistributedMLP, td_mlp_forward, upsample_linear
from .modules.generators import FIRNoiseSynth, HarmonicOscillator
from .modules.shaping import NEWT, Reverb

gin.external_configurable(nn.GRU, module="torch.nn")
gin.external_configurable(nn.Conv1d, module="torch.nn")

_DEFAULT_GIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gin", "models", "newt.gin")


@gin.configurable
class ControlModule(nn.Module):
    """GRU(control_size -> hidden) + Conv1d(hidden -> embedding, 1) (reference :17-26)."""

    def __init__(self, control_size: int, hidden_size: int, embedding_size: int):
        super().__init__()
        self.gru = nn.GRU(control_size, hidden_size, batch_first=True)
        self.proj = nn.Conv1d(hidden_size, embedding_size, 1)

        self._desc = sa.Desc()

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_desc"] = sa.Desc()
        return d

    def forward(self, x):
        """(B, control_size, T) -> (B, embedding_size, T): persistent GRU kernel (csrc/control_gru.hip) + Conv1d(k=1).
        Stand-alone form of what NeuralWaveshaping.forward runs fused (GRU, then proj inside frame_mlps16_kernel)."""
        import ctypes as C

        x = sa.contiguous(x, "x")
        g = self.gru
        if g.num_layers != 1 or g.bidirectional or not g.batch_first:
            raise RuntimeError("the HIP path implements nn.GRU(control_size, hidden_size, batch_first=True), one layer")
        if x.dim() != 3 or x.shape[1] != g.input_size:
            raise RuntimeError(f"ControlModule: expected (B, {g.input_size}, T), got {tuple(x.shape)}")
        if g.input_size != 2 or g.hidden_size != sa._lib.HIDDEN:
            # any control_size / hidden_size: runtime-size recurrence (csrc/generic.hip: g_gru_kernel)
            ps = [sa._req(p.detach(), "gru parameter") for p in (g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)]
            sa.no_autograd(params=[g.weight_ih_l0])
            B, Cin, T = x.shape
            H = g.hidden_size

            def g_call(lib):
                with torch.cuda.device(x.device):
                    out = torch.empty((B, T, H), dtype=torch.float32, device=x.device)
                    nb = lib.nws_g_gru_workspace_bytes(H)
                    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
                    sa.checked(lib.nws_g_gru(ps[0].data_ptr(), ps[1].data_ptr(), ps[2].data_ptr(), ps[3].data_ptr(), x.data_ptr(), B,
                                             Cin, Cin, H, T, None, out.data_ptr(), None, ws.data_ptr(), nb, sa.stream_ptr(x.device)),
                               "nws_g_gru")
                return out

            o = sa.ops()
            sa.no_autograd(inputs=(x,))
            h = o.g_gru(ps[0], ps[1], ps[2], ps[3], x, None)[0] if o is not None else g_call(sa._lib.lib())
            return td_mlp_forward(h.transpose(1, 2).contiguous(), self.proj)
        w, _, wdesc = self._desc.get({"gru_w_ih": g.weight_ih_l0, "gru_w_hh": g.weight_hh_l0, "gru_b_ih": g.bias_ih_l0,
                                      "gru_b_hh": g.bias_hh_l0})

        def c_call(L):
            with torch.cuda.device(x.device):
                out = torch.empty((x.shape[0], x.shape[2], sa._lib.HIDDEN), dtype=torch.float32, device=x.device)
                sa.checked(L.nws_control_gru(C.byref(w), x.data_ptr(), x.shape[0], x.shape[1], x.shape[2], out.data_ptr(),
                                             sa.stream_ptr(x.device)), "nws_control_gru")
            return out

        o = sa.ops()
        h = o.control_gru(wdesc, x, None, False)[0] if o is not None else c_call(sa._lib.lib())      # (B, T, 128)
        return td_mlp_forward(h.transpose(1, 2).contiguous(), self.proj)


@gin.configurable
class NeuralWaveshaping(nn.Module):
    def __init__(self, n_waveshapers: int, control_hop: int, sample_rate: float = 16000,
                 learning_rate: float = 1e-3, lr_decay: float = 0.9, lr_decay_interval: int = 10000,
                 log_audio: bool = False):
        super().__init__()
        self.hparams = dict(n_waveshapers=n_waveshapers, control_hop=control_hop, sample_rate=sample_rate,
                            learning_rate=learning_rate, lr_decay=lr_decay, lr_decay_interval=lr_decay_interval,
                            log_audio=log_audio)
        self.learning_rate = learning_rate
        self.lr_decay = lr_decay
        self.lr_decay_interval = lr_decay_interval
        self.control_hop = control_hop
        self.log_audio = log_audio
        self.sample_rate = sample_rate

        self.embedding = ControlModule()
        self.osc = HarmonicOscillator()
        self.harmonic_mixer = Conv1x1(self.osc.n_harmonics, n_waveshapers, 1)
        self.newt = NEWT()
        with gin.config_scope("noise_synth"):
            self.h_generator = TimeDistributedMLP()
            self.noise_synth = FIRNoiseSynth()
        self.reverb = Reverb()
        object.__setattr__(self, "_engine", Engine(self))

    # ---- cache hygiene: any re-homing / re-loading of parameters drops the pointer cache ----------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine.invalidate()
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._engine.invalidate()
        return out

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        if name == "newt" and "_engine" in self.__dict__:
            self._engine.invalidate()

    def invalidate_cache(self):
        """Drop the engine's cached pointer struct and derived tables.  In-place parameter updates are noticed on their own
        (Engine._fingerprint); needed only after writes through `p.data` or after changing `exciter_opts`."""
        self._engine.invalidate()

    # a copy / an unpickled model gets its own engine (the cache holds raw device pointers of THIS model's tensors)
    def __getstate__(self):
        d = self.__dict__.copy()
        d.pop("_engine", None)
        return d

    def __setstate__(self, state):
        super().__setstate__(state)
        object.__setattr__(self, "_engine", Engine(self))

    # ---- public surface ----------------------------------------------------------------------------
    def render_exciter(self, f0):
        """(B, 1, N) upsampled F0 in Hz -> (B, n_waveshapers, N) exciter (reference :64-67)."""
        f0 = _req(f0 if f0.is_contiguous() else f0.contiguous(), "f0")
        if f0.dim() != 3 or f0.shape[1] != 1:
            raise RuntimeError(f"expected (B, 1, N), got {tuple(f0.shape)}")
        if not self._engine.specialised() or f0.shape[-1] % _lib.HOP:
            # any n_harmonics / n_waveshapers / length: oscillator bank + Conv1d(k=1), runtime-size kernels (csrc/generic.hip)
            with torch.no_grad():
                osc = self.osc(f0[:, 0])
                mw = _req(self.harmonic_mixer.weight.detach(), "harmonic_mixer.weight")
                mb = _req(self.harmonic_mixer.bias.detach(), "harmonic_mixer.bias")
                B, K, N = osc.shape
                S = mw.shape[0]

                def c_call(lib):
                    with torch.cuda.device(osc.device):
                        out = torch.empty((B, S, N), dtype=torch.float32, device=osc.device)
                        sa.checked(lib.nws_g_conv1x1(osc.data_ptr(), mw.data_ptr(), mb.data_ptr(), B, K, S, N, out.data_ptr(),
                                                     sa.stream_ptr(osc.device)), "nws_g_conv1x1")
                    return out

                return sa.call("g_conv1x1", "nws_g_conv1x1", (osc, mw, mb), c_call)
        eng = self._engine
        f0_up = f0[:, 0]
        u = torch.rand_like(self.osc.rand_phase).reshape(-1)
        carry = eng.phase_carry(f0_up=f0_up)
        exc, _ = eng.exciter_newt(None, f0_up.contiguous(), carry, u, None, want_exciter=True, want_newt=False)
        return exc

    def get_embedding(self, control):
        """(B, C>=2, T) normalised control -> (B, 128, T) embedding (reference :69-72)."""
        control = _req(control if control.is_contiguous() else control.contiguous(), "control")
        if not self._engine.specialised():
            return self.embedding(control[:, 0:2].contiguous())       # (reference :70-72: f0 and loudness channels only)
        gru = self._engine.control_gru(control)
        emb, _, _, _ = self._engine.frame_mlps(gru, want_emb=True)
        return emb

    def forward(self, f0, control, *, phase_u=None, noise=None):
        # strided views (control[:, :2], expand()) are accepted like in the reference: compacted by torch
        f0 = _req(f0 if f0.is_contiguous() else f0.contiguous(), "f0")
        control = _req(control if control.is_contiguous() else control.contiguous(), "control")
        if f0.dim() != 3 or f0.shape[1] != 1:
            raise RuntimeError(f"f0: expected (B, 1, T), got {tuple(f0.shape)}")
        if control.dim() != 3 or control.shape[1] < 2:
            raise RuntimeError(f"control: expected (B, C>=2, T), got {tuple(control.shape)}")
        B, _, T = f0.shape
        if control.shape[0] != B or control.shape[2] != T:
            raise RuntimeError(f"f0 {tuple(f0.shape)} and control {tuple(control.shape)} disagree on batch / frames")
        if T < 2:
            raise RuntimeError("need at least 2 control frames (reflect padding of the noise STFT, generators.py:31)")
        dev = f0.device
        sa.no_autograd(inputs=(f0, control))     # inference-only kernels: never hand back a graph-less result for grad inputs
        if phase_u is None:
            phase_u = torch.rand_like(self.osc.rand_phase)          # RNG draw #1 (generators.py:55)
        phase_u = _req(phase_u.reshape(-1), "phase_u", int(self.osc.n_harmonics))
        if noise is None:
            noise = torch.rand(self.control_hop * T - 1, device=dev)  # RNG draw #2 (generators.py:30)
        noise = _req(noise, "noise", self.control_hop * T - 1)
        if self._sub_module_hooks():
            return self._forward_module_by_module(f0, control, phase_u, noise)
        return self._engine.forward(f0, control, phase_u, noise)

    def _sub_module_hooks(self) -> bool:
        """Forward hooks on sub-modules (the reference's users tap stages that way, and so does tests/golden/make_golden.py on
        the reference itself): the fused forward never calls the sub-modules, so with hooks present the forward runs them one
        by one instead - the reference's own sequence (models/neural_waveshaping.py:74-90), one HIP stage kernel per module."""
        hit = self.__dict__.get("_hook_mods")
        if hit is None or hit[0] != _engine_epoch():
            # the hook registries themselves (one OrderedDict per module and kind, created once in Module.__init__): checking
            # ~90 dicts for emptiness is ~2 us per forward
            hit = (_engine_epoch(), [d for m in self.modules() if m is not self for d in (m._forward_hooks, m._forward_pre_hooks)])
            self.__dict__["_hook_mods"] = hit
        for d in hit[1]:
            if d:
                return True
        return False

    def _forward_module_by_module(self, f0, control, phase_u, noise):
        with torch.no_grad():
            f0_up = upsample_linear(f0, int(self.control_hop))                    # :75
            sig = self.osc(f0_up[:, 0], phase_u=phase_u)                          # :65   (draw #1 injected / made above)
            x = self.harmonic_mixer(sig)                                          # :66
            emb = self.embedding(control[:, 0:2].contiguous())                    # :69-72
            x = self.newt(x, emb)                                                 # :80
            H = self.h_generator(emb)                                             # :82
            nz = self.noise_synth(H, noise=noise)                                 # :83   (draw #2)
            x = sa.sum_channels(torch.cat((x, nz), dim=1))                        # :85-86 (cat: plumbing; the sum: a HIP kernel)
            return self.reverb(x)                                                 # :88

    # ---- checkpoints (Lightning .ckpt as shipped by the reference, or flat .npz fixtures) -------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **kwargs):
        from ..checkpoint import read_checkpoint

        ensure_default_config()
        state, hparams = read_checkpoint(checkpoint_path)
        hparams.update(kwargs)
        model = cls(**hparams)
        model.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()}, strict=strict)
        if map_location is not None:
            model = model.to(map_location)
        return model


def _engine_epoch():
    from ..engine import _EPOCH

    return _EPOCH[0]


def ensure_default_config():
    """Parse the packaged newt.gin if the caller has not parsed a model gin file yet."""
    try:
        gin.query_parameter("NeuralWaveshaping.control_hop")
    except ValueError:
        gin.parse_config_file(_DEFAULT_GIN)


def _stream(self, batch_size: int = 1, *, phase_u=None, noise=None, **kw):
    """Stateful streaming synthesiser bound to this model (see streaming.NewtStream; kw: max_chunk_frames, graph)."""
    from ..streaming import NewtStream

    return NewtStream(self, batch_size, phase_u=phase_u, noise=noise, **kw)


NeuralWaveshaping.stream = _stream
