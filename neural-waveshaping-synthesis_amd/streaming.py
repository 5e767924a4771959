"""Stateful streaming synthesis on top of the one-shot HIP kernels.

The reference's buffer benchmark (scripts/time_buffer_sizes.py) is stateless: every buffer restarts the GRU, the
oscillator phase and the reverb.  `NewtStream` carries that state so that the concatenation of the chunks it emits
equals the reference's ONE-SHOT forward over the whole signal up to the reverb input (`pre_reverb`), and applies the
learned reverb as a linear (overlap-add) convolution instead of the one-shot path's wrap-around.

How the one-shot kernels are reused (no streaming-specific sample-rate kernels):
  * control frames arrive in chunks of K frames; every kernel runs on a WINDOW = [last frame of the previous chunk] +
    [K new frames].  The linear upsampling of F0 / FiLM parameters only looks one frame back and one ahead, so all
    window samples except the first and last 64 are exactly the one-shot values; the stream therefore emits audio
    64 samples (4 ms) behind the control frames it has seen (`push(..., final=True)` releases the remainder).
  * GRU hidden state: carried (nws_control_gru_state).
  * oscillator phase: the float64 prefix sum of the upsampled F0 is carried as one double per utterance and spliced
    into the per-window carries, so `fl32(cumsum)` is bit-identical to the one-shot forward's.
  * noise branch: frames are cut from one absolute noise stream (nws_fir_noise_window with origin 0); the branch runs
    64 samples ahead of the NEWT branch, the surplus is kept as a residue for the next chunk.
  * reverb: nws_reverb_linear_chunk with a 32000-sample tail per utterance.
The phase offsets are drawn once per stream and the noise is drawn chunk by chunk from the device generator (or both
are injected for parity testing), mirroring the reference's two hidden draws.
"""
from __future__ import annotations

import torch

from . import _lib
from .engine import _req

HOP = _lib.HOP
_MAX_CHUNK_FRAMES = 249   # (K+1) * 128 + 31999 <= 64000: one linear-reverb chunk


class NewtStream:
    def __init__(self, model, batch_size: int, phase_u: torch.Tensor | None = None, noise: torch.Tensor | None = None):
        self.model = model
        self.eng = model._engine
        w, _, dev = self.eng.weights()
        self.dev = dev
        self.B = int(batch_size)
        self.phase_u = _req((phase_u if phase_u is not None else torch.rand_like(model.osc.rand_phase)).reshape(-1),
                            "phase_u", _lib.N_HARMONICS)
        self._noise_fixed = noise is not None          # parity mode: the reference's whole draw is supplied up front
        self._noise = _req(noise, "noise") if noise is not None else torch.empty(0, dtype=torch.float32, device=dev)
        self._noise_base = 0                            # absolute index of self._noise[0]
        self.h = torch.zeros((self.B, _lib.HIDDEN), dtype=torch.float32, device=dev)
        self.prev = None                                # (f0 (B,1), film (B,1,256), fir (B,1,256)) of the last frame seen
        self.S = torch.zeros((self.B, 1), dtype=torch.float64, device=dev)   # fp64 phase sum through the last emitted sample
        self.noise_residue = torch.zeros((self.B, 0), dtype=torch.float32, device=dev)
        self.frames_seen = 0
        self.samples_emitted = 0
        self.finished = False
        ir = self.eng.ir()
        self.tail_len = ir.numel() + 1
        self._rv_aux = self.eng.reverb_aux(2 * self.tail_len)   # L = 64000 >= chunk + 32000 - 1
        self.plan = self._rv_aux[0]
        self.tails = [torch.zeros((self.B, self.tail_len), dtype=torch.float32, device=dev) for _ in range(2)]
        self._tail_idx = 0

    # ---- noise stream ---------------------------------------------------------------------------------------------
    def _noise_view(self, start: int, upto: int):
        """View of the absolute noise stream beginning at `start`, guaranteed to hold indices < upto (unless fixed)."""
        if not self._noise_fixed:
            have = self._noise_base + self._noise.numel()
            if upto > have:
                extra = torch.rand(max(upto - have, 4096), device=self.dev)
                self._noise = torch.cat((self._noise, extra))
            drop = start - self._noise_base - 1024                      # forget what no future window can reach
            if drop > 65536:
                self._noise = self._noise[drop:].contiguous()
                self._noise_base += drop
        return self._noise[start - self._noise_base:]

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
        if K > _MAX_CHUNK_FRAMES:                       # long chunks: process in pieces (state makes that exact)
            outs = []
            for k0 in range(0, K, _MAX_CHUNK_FRAMES):
                k1 = min(K, k0 + _MAX_CHUNK_FRAMES)
                outs.append(self.push(f0[:, :, k0:k1], control[:, :, k0:k1], final and k1 == K))
            return torch.cat(outs, dim=1)
        eng = self.eng
        first = self.prev is None

        # 1. control path on the K new frames (GRU state carried)
        gru, self.h = eng.control_gru(control, h0=self.h, return_state=True)
        _, film, _, fir = eng.frame_mlps(gru)

        # 2. window = [previous frame] + new frames
        f0_new = f0[:, 0, :]
        if first:
            f0_w, film_w, fir_w = f0_new.contiguous(), film, fir
        else:
            f0_w = torch.cat((self.prev[0], f0_new), dim=1).contiguous()
            film_w = torch.cat((self.prev[1], film), dim=1).contiguous()
            fir_w = torch.cat((self.prev[2], fir), dim=1).contiguous()
        Tw = f0_w.shape[1]
        lo = 0 if first else 64
        hi = HOP * Tw if final else HOP * Tw - 64

        # 3. exciter + waveshapers on the window, with the carried fp64 phase sum spliced into the chunk carries
        carry = eng.phase_carry(f0=f0_w)                               # (B, 4 Tw) exclusive prefix sums, float64
        if not first:
            carry = self.S + (carry - carry[:, 2:3])                   # exact: sums of fp32 values in fp64
        # the kernel's lerp clamps to the window edges; those samples are only kept at the true stream edges
        _, newt = eng.exciter_newt(f0_w, None, carry.contiguous(), self.phase_u, film_w)
        new_S = None if final else carry[:, (hi // 32):(hi // 32) + 1].clone()
        newt_emit = newt[:, lo:hi]

        # 4. noise branch on the same window, from the absolute noise stream
        A0 = self.frames_seen - (0 if first else 1)                   # absolute index of the window's first frame
        if A0 == 0:
            start, origin = 0, HOP            # frame 0 reaches 128 samples before the stream: reflect like torch.stft
        else:
            start, origin = HOP * A0 - HOP, 0
        need_upto = HOP * (A0 + Tw - 1) + HOP + 1
        nview = self._noise_view(start, need_upto)
        if final and self._noise_fixed:
            n_len = (HOP * (A0 + Tw) - 1) - start                    # the one-shot draw has N-1 samples: reflect at its end
        elif final:
            n_len = (HOP * (A0 + Tw) - 1) - start
            nview = self._noise_view(start, start + n_len)
        else:
            n_len = nview.numel()
        if n_len > nview.numel():
            raise RuntimeError("injected noise vector is shorter than the stream (needs 128*frames - 1 samples)")
        noise_w = eng.fir_noise(fir_w, nview, origin=origin, noise_len=int(n_len))
        noise_new = noise_w if first else noise_w[:, HOP:]             # local hop 0 of a non-first window is not ours
        noise_all = torch.cat((self.noise_residue, noise_new), dim=1)
        cnt = hi - lo
        pre = (newt_emit + noise_all[:, :cnt]).contiguous()
        self.noise_residue = noise_all[:, cnt:].contiguous()

        # 5. linear reverb with carried tail
        # (fetched per push: cached by the engine per weights version, so an in-place update of reverb.ir is picked up)
        self._rv_aux = eng.reverb_aux(2 * self.tail_len)
        y, self.tails[self._tail_idx ^ 1] = eng.reverb_linear_chunk(self._rv_aux, pre, self.tails[self._tail_idx])
        self._tail_idx ^= 1

        # 6. state for the next chunk
        self.prev = (f0_new[:, -1:].contiguous(), film[:, -1:, :].contiguous(), fir[:, -1:, :].contiguous())
        if new_S is not None:
            self.S = new_S
        self.frames_seen += K
        self.samples_emitted += cnt
        self.finished = final
        self._last_pre = pre
        return y

    def reverb_tail(self) -> torch.Tensor:
        """The remaining (B, 32000) reverb tail after the last chunk (what a linear reverb still rings out)."""
        return self.tails[self._tail_idx].clone()
