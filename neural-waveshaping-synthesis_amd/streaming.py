"""Stateful streaming synthesis: host side of csrc/stream.hip.

The reference's buffer benchmark (scripts/time_buffer_sizes.py) is stateless: every buffer restarts the GRU, the
oscillator phase and the reverb.  `NewtStream` carries that state so that the concatenation of the chunks it emits
equals the reference's ONE-SHOT forward over the whole signal up to the reverb input (`pre_reverb`), and applies the
learned reverb as a linear convolution of the stream instead of the one-shot path's wrap-around.

One `push` = ONE C-ABI call (`nws_stream_step` / `torch.ops.newt_hip.stream_step`): the one-shot kernels plus small streaming
kernels on a window = [last frame of the previous chunk] + [K new frames] - four launches for a hop of <= 256 samples (whatever
depends on nothing the hop computes, and the frame MLPs of its one or two frames, ride on the recurrence launch: DESIGN.md 3.8),
seven for longer chunks; every piece of state
(GRU h, previous frame, float64 phase sum, noise residue and window, reverb-input ring, position counters) lives in one
device blob behind fixed pointers.  Nothing is computed by torch.  Because the pointers and launch arguments of a
steady-state hop (same K as the previous push, not first, not final) never change, such hops are captured ONCE into a
hipGraph and replayed (`graph=True`, the default): a 256-sample hop is then one graph launch.

  * the stream emits audio 64 samples (4 ms) behind the control frames it has seen (linear upsampling looks one frame
    ahead); `push(..., final=True)` releases the remainder exactly like the one-shot forward's right edge;
  * the phase offsets are drawn once per stream and the noise chunk by chunk from the device generator (or both are
    injected for parity testing), mirroring the reference's two hidden draws.
"""
from __future__ import annotations

import ctypes as C
import time

import torch

from . import _lib
from .engine import _req, ops, stream_ptr

HOP = _lib.HOP
_MAX_CHUNK_FRAMES = 249   # 16 kHz: (K+1) * 128 + 31999 <= 64000, the FFT reverb of a long chunk inside the L = 64000 plan
_RING = 65536             # kRing of csrc/stream.hip: reverb-input ring per utterance; the impulse response must fit half of it
_GRAPH_AFTER = 2          # consecutive steady-state pushes of one (K, channels) before that hop is captured


class NewtStream:
    def __init__(self, model, batch_size: int, phase_u: torch.Tensor | None = None, noise: torch.Tensor | None = None,
                 max_chunk_frames: int = _MAX_CHUNK_FRAMES, graph: bool = True):
        if not model._engine.specialised():
            raise RuntimeError("stateful streaming runs on the fused kernels: the model must have the architecture of "
                               "gin/models/newt.gin")
        self.model = model
        self.eng = model._engine
        w, _, dev = self.eng.weights()
        self.dev = dev
        self.B = int(batch_size)
        self._ir_len = int(self.eng.ir().numel())
        self.tail_len = self._ir_len + 1
        if self._ir_len >= _RING // 2:
            raise RuntimeError(f"stateful streaming keeps {_RING // 2 - 1} samples of reverb history per utterance; this model's impulse "
                               f"response has {self._ir_len} taps ({self._ir_len / float(model.sample_rate):.2f} s at "
                               f"{model.sample_rate} Hz) - render it with the one-shot forward")
        # the FFT reverb of a long chunk is one direct transform of [ir_len samples of history | chunk]: the smallest standard
        # length that holds twice the tail (64000 at 16 kHz, 32000 at 8 kHz) bounds the chunk
        self._plan_n = next(n for n in (32000, 64000, 128000, 256000) if n >= 2 * self.tail_len)
        fit = (self._plan_n - self._ir_len) // HOP - 1
        self.max_frames = int(min(max(1, max_chunk_frames), fit))
        self.phase_u = _req((phase_u if phase_u is not None else torch.rand_like(model.osc.rand_phase)).reshape(-1),
                            "phase_u", _lib.N_HARMONICS)
        self._noise_all = _req(noise, "noise") if noise is not None else None    # parity mode: the reference's whole draw
        self.frames_seen = 0
        self.samples_emitted = 0
        self.finished = False
        self._nz_prev_start = 0
        self._need_fft = HOP * (self.max_frames + 1) > 2048
        L = _lib.lib()
        plan = self.eng.reverb_aux(self._plan_n)[0] if self._need_fft else None
        nbytes = L.nws_stream_state_bytes(self.B, self.max_frames, self._ir_len, C.byref(plan) if plan is not None else None)
        with torch.cuda.device(dev):
            self._state = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.nws_stream_reset(self._state.data_ptr(), nbytes, stream_ptr(dev)), "nws_stream_reset")
        self._use_graph = bool(graph)
        self._graphs = {}          # (K, channels) -> (graph, f0_in, control_in, noise_new, out, pre)
        self._steady_runs = {}     # (K, channels) -> CONSECUTIVE steady-state pushes seen
        self._last_K = None
        self._last_pre = None

    # ---- one C-ABI call ---------------------------------------------------------------------------------------------
    def _step(self, f0_2d, control, first, final, noise_new, out, pre):
        eng = self.eng
        w, keep, dev, wdesc = eng._wd()
        B, K = f0_2d.shape
        fft = self._need_fft
        plan, tables, spec, plan_t = eng._reverb_aux(self._plan_n) if fft else (None, None, None, None)
        sr = eng.osc_sample_rate()
        rp, ir = keep[-2], keep[-1]
        o = ops()
        if o is not None:
            o.stream_step(wdesc, eng._fir_design, plan_t, tables, spec, self._state, self.max_frames, f0_2d, control, bool(first),
                          bool(final), int(self.frames_seen), int(self._nz_prev_start), sr, self.phase_u, rp, noise_new,
                          self._noise_all, ir.reshape(-1), out, pre)
            return
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().nws_stream_step(
                C.byref(w), eng._fir_design.data_ptr(), C.byref(plan) if fft else None, tables.data_ptr() if fft else None,
                spec.data_ptr() if fft else None, self._state.data_ptr(), self._state.numel(), B, self.max_frames,
                f0_2d.data_ptr(), control.data_ptr(), control.shape[1], K, int(first), int(final), int(self.frames_seen),
                int(self._nz_prev_start), sr, self.phase_u.data_ptr(), rp.data_ptr(),
                noise_new.data_ptr() if noise_new is not None else None,
                self._noise_all.data_ptr() if self._noise_all is not None else None,
                self._noise_all.numel() if self._noise_all is not None else 0, ir.data_ptr(), ir.numel(), out.data_ptr(),
                pre.data_ptr() if pre is not None else None, stream_ptr(dev)), "nws_stream_step")

    def _capture(self, K, C_in):
        """hipGraph of one steady-state hop of K frames: static input / output buffers, the fresh noise draws inside."""
        dev = self.dev
        with torch.cuda.device(dev):
            f0_in = torch.zeros((self.B, K), dtype=torch.float32, device=dev)
            c_in = torch.zeros((self.B, C_in, K), dtype=torch.float32, device=dev)
            nz = torch.empty(HOP * K, dtype=torch.float32, device=dev) if self._noise_all is None else None
            out = torch.empty((self.B, HOP * K), dtype=torch.float32, device=dev)
            pre = torch.empty((self.B, HOP * K), dtype=torch.float32, device=dev)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(g):
                if nz is not None:
                    nz.uniform_()                                  # the reference's torch.rand draw, chunk by chunk
                self._step(f0_in, c_in, False, False, nz, out, pre)
        # capture records, it does not run: the state has not advanced
        return g, f0_in, c_in, nz, out, pre

    # ---- one chunk ----------------------------------------------------------------------------------------------------
    def push(self, f0: torch.Tensor, control: torch.Tensor, final: bool = False) -> torch.Tensor:
        """f0 (B,1,K) Hz, control (B,C>=2,K) normalised  ->  (B, 128K) audio (128K-64 for the first chunk, 128K+64
        for the chunk marked `final`, which ends the stream exactly like the one-shot forward's right edge)."""
        if self.finished:
            raise RuntimeError("stream already finished")
        f0 = _req(f0 if f0.is_contiguous() else f0.contiguous(), "f0")
        control = _req(control if control.is_contiguous() else control.contiguous(), "control")
        B, _, K = f0.shape
        if B != self.B or control.shape[0] != B or control.shape[2] != K or control.shape[1] < 2:
            raise RuntimeError(f"expected f0 ({self.B},1,K) and control ({self.B},C>=2,K)")
        if K < 1:
            raise RuntimeError("empty chunk")
        if K > self.max_frames:                          # long chunks: process in pieces (state makes that exact)
            outs = []
            for k0 in range(0, K, self.max_frames):
                k1 = min(K, k0 + self.max_frames)
                outs.append(self.push(f0[:, :, k0:k1], control[:, :, k0:k1], final and k1 == K))
            return torch.cat(outs, dim=1)
        first = self.frames_seen == 0
        L = _lib.lib()
        M = L.nws_stream_out_samples(K, int(first), int(final))
        if M <= 0:
            raise RuntimeError("a first chunk of one frame that is also not final emits 64 samples; nothing smaller exists")
        steady = (not first) and (not final) and self._last_K == K and self.frames_seen >= K + 2
        f0_2d = f0[:, 0, :]
        self._check_weights()
        key = (K, control.shape[1])
        if not steady:
            self._steady_runs.clear()      # the count is of CONSECUTIVE steady hops of one shape
        if steady and self._use_graph:
            hit = self._graphs.get(key)
            runs = self._steady_runs.get(key, 0) + 1
            self._steady_runs = {key: runs}
            if hit is None and runs > _GRAPH_AFTER and not torch.cuda.is_current_stream_capturing():
                try:
                    hit = self._graphs[key] = self._capture(K, control.shape[1])
                except Exception as e:     # capture not possible here (e.g. foreign capture in progress): stay eager, and say so
                    import warnings
                    warnings.warn(f"NewtStream: hipGraph capture of the {K}-frame hop failed ({type(e).__name__}: {e}); this stream "
                                  f"continues with eager pushes", RuntimeWarning, stacklevel=2)
                    self._use_graph = False
                    hit = None
            if hit is not None:
                g, f0_in, c_in, _, out, pre = hit
                torch._foreach_copy_([f0_in, c_in], [f0_2d, control])      # one multi-tensor launch for both inputs
                g.replay()
                self._advance(K, M, first, final)
                self._last_pre = pre
                return out.clone()
        with torch.cuda.device(self.dev):
            out = torch.empty((B, M), dtype=torch.float32, device=self.dev)
            pre = torch.empty((B, M), dtype=torch.float32, device=self.dev)
            nz = None
            if self._noise_all is None:
                nz = torch.rand(L.nws_stream_noise_draws(K, int(first), self.frames_seen), device=self.dev)   # RNG draw #2, chunk-wise
            self._step(f0_2d.contiguous(), control, first, final, nz, out, pre)
        self._advance(K, M, first, final)
        self._last_pre = pre
        return out

    def refresh(self):
        """Pick up a weight update NOW: re-derive the engine's tables if any parameter changed and drop the captured hops that
        point into the old ones.  Call it after an optimizer step / load_state_dict / in-place edit when the very next hop must
        see the new weights; without it a captured hop (`graph=True`) keeps replaying the old tables for up to 250 ms (eager
        pushes notice at once), see _check_weights."""
        self.__dict__["_last_walk"] = 0.0
        self._check_weights()

    def _check_weights(self):
        """A captured hop holds raw pointers into the engine's derived tables (fragment tables, LUT pairs, FIR design, IR
        spectrum).  If the engine has rebuilt them (somebody ran a forward after a weight update, `.to()`, `invalidate_cache`)
        the graphs are dropped and re-captured; in-place updates nobody has told the engine about are looked for at most every
        250 ms of wall-clock (a full fingerprint walk costs ~12 us of host time: too much for every 256-sample hop, nothing once
        per sixteen 16 ms hops).  So for up to 250 ms after such an update graph=True and graph=False streams differ; `refresh()`
        closes that window on demand."""
        eng = self.eng
        now = time.monotonic()
        if now - self.__dict__.get("_last_walk", 0.0) >= 0.25:
            self._last_walk = now
            if eng._w is not None and eng._fingerprint() != eng._fp:
                eng._wd()                              # rebuilds (drains the device first)
        if eng._w is not self.__dict__.get("_w_seen"):
            if self._graphs:
                torch.cuda.synchronize(self.dev)
                self._graphs.clear()
                self._steady_runs.clear()
            self._w_seen = eng._w

    # ---- zero-copy hops: the caller writes into the captured hop's own input buffers and reads its output buffer ----------
    def static_io(self, K: int, channels: int = 2):
        """(f0_in (B, K), control_in (B, channels, K), out (B, 128 K)) of the captured steady-state hop of K frames - the
        buffers `hop()` consumes and fills, for callers that produce control frames in place (an audio callback): no input
        copies, no output copy.  Needs the stream in steady state for this K (two pushes of K frames behind it)."""
        if not (self.frames_seen >= K + 2 and self._last_K == K and not self.finished):
            raise RuntimeError(f"static_io({K}): push at least two chunks of {K} frames first (the captured hop is the steady-state one)")
        hit = self._graphs.get((K, channels))
        if hit is None:
            hit = self._graphs[(K, channels)] = self._capture(K, channels)
        return hit[1], hit[2], hit[4]

    def hop(self, K: int, channels: int = 2) -> torch.Tensor:
        """Replay the captured hop on whatever the caller left in static_io(K)'s input buffers; returns the static output
        buffer (overwritten by the next hop)."""
        self._check_weights()
        hit = self._graphs.get((K, channels))
        if hit is None or self._last_K != K or self.finished or self.frames_seen < K + 2:
            raise RuntimeError("hop(): call static_io(K) first (again after a weight update), and do not interleave other chunk sizes")
        hit[0].replay()
        self._advance(K, HOP * K, False, False)
        self._last_pre = hit[5]
        return hit[4]

    def _advance(self, K, M, first, final):
        self._nz_prev_start = int(_lib.lib().nws_stream_noise_start(int(first), self.frames_seen))
        self.frames_seen += K
        self.samples_emitted += M
        self._last_K = K
        self.finished = bool(final)

    def reverb_tail(self) -> torch.Tensor:
        """The remaining (B, 32000) reverb tail after the last chunk (what a linear reverb still rings out)."""
        eng = self.eng
        eng._wd()
        plan, tables, spec, _ = eng._reverb_aux(self._plan_n)
        L = _lib.lib()
        with torch.cuda.device(self.dev):
            tail = torch.empty((self.B, self.tail_len), dtype=torch.float32, device=self.dev)
            nb = 2 * ((self.B * (2 * self._ir_len + 1) * 4 + 255) // 256 * 256) + L.nws_reverb_workspace_bytes(C.byref(plan), self.B)
            ws = torch.empty(nb, dtype=torch.uint8, device=self.dev)
            _lib.check(L.nws_stream_reverb_tail(C.byref(plan), tables.data_ptr(), spec.data_ptr(), self._state.data_ptr(),
                                                self._state.numel(), self.B, self.max_frames, self._ir_len, tail.data_ptr(),
                                                ws.data_ptr(), nb, stream_ptr(self.dev)), "nws_stream_reverb_tail")
        return tail
