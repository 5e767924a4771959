"""Throughput mode of NeuralWaveshaping.forward: successive batches pipelined over HIP streams.

One forward (models/neural_waveshaping.py:74-90) is a chain  carries -> GRU -> frame MLPs -> oscillator+NEWT -> noise ->
reverb.  Everything but the GRU fills all 256 CUs; the GRU is a 500-step recurrence that is latency-bound whatever the
batch size (B workgroups).  Whole forwards issued round-robin on several streams fall into lock-step (all streams in their
GRU at once, then all in their oscillator).  `ForwardPipeline` instead issues the *control half* (carries + GRU) of batch
i+1 on a side stream while the *audio half* of batch i occupies the GPU, with events for the hand-over and a small ring of
workspaces: 0.76 ms (plain forwards) -> 0.51 (one audio stream) -> 0.46 ms (two) per 64 x 4 s batch on MI355X.  The kernels
and their results are exactly those of `model(f0, control)`; only the issue order across batches changes.

    pipe = ForwardPipeline(model)
    outs = [pipe.submit(f0_i, control_i) for ...]     # asynchronous; draws the reference's two RNG vectors per batch
    pipe.synchronize()                                 # or pipe.join_current_stream() to stay asynchronous

Inputs must be valid in the submitting thread's current stream at submit() time (an event is recorded there and the
internal streams wait for it).

Control streams.  One is the default.  When the audio half is short (realistic F0: the oscillator skips the harmonics above
Nyquist, 0.19 instead of 0.31 ms) the GRU of the next batch becomes the longer half and `control_streams=2` lets two of them
overlap: 0.392 -> 0.364 ms per batch on one GPU; next to RCCL's own streams it was slower (0.45 vs 0.39), hence not the default.

Audio streams.  `audio_streams=2` alternates the audio halves over two streams (another ~6 %: the tail of batch i - noise,
reverb - overlaps the head of batch i+1).  That configuration exposed a hardware hazard on MI355X which the build now guards
against (LABBOOK.md '5.2', csrc/coexec_probe.hip): a packed fp32 instruction whose low lane reads the high half of its
second operand (v_pk_{add,mul,fma}_f32 with op_sel[1] = 1 - what a complex "times -i" butterfly compiles to) returns wrong
values while ANOTHER kernel executes K=16/32 f16 MFMAs on the same compute unit.  The reverb's FFT kernels of batch i, running
beside the frame-MLP / noise kernels of batch i+1, came out wrong in pairs of rows (~1e-2) in up to half of the batches.  No
product kernel contains that instruction form any more (neural-waveshaping-synthesis_amd/build.py fails the build if one does), and
tests/test_gpu_coexec.py soaks exactly this configuration bit for bit.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from .engine import _req


class _Slot:
    __slots__ = ("ws", "ev_control", "ev_audio", "ev_exciter", "used", "keep", "pu", "nz")

    def __init__(self):
        self.ws = None
        self.pu = None          # this slot's own buffers for the two hidden draws (filled on the control stream)
        self.nz = None
        self.ev_control = torch.cuda.Event()
        self.ev_audio = torch.cuda.Event()
        self.ev_exciter = torch.cuda.Event()       # recorded right after this batch's oscillator + waveshaper kernel
        self.used = False
        self.keep = None


_STREAM_SETS = {}


def placed_streams(dev, audio_streams: int = 2, control_streams: int = 2):
    """(exchange stream, audio streams, control streams) of a device, created ONCE per process and shape and put on the command
    processor's pipes in a fixed pattern.

    Measured on MI355X (round 5, `tools/queue_order_probe.sh`, profiles/r05/queue_placement.txt): HIP creates a stream's
    hardware queue at the stream's FIRST use, and the k-th hardware queue of a process (all priorities counted together) is
    served by pipe k % 4 of the command processor.  A pipe dispatches one kernel at a time, and the oscillator kernel keeps its
    pipe busy for its whole duration (16 000 workgroups are handed out as slots free up), so
      * the two audio and two control streams must sit on FOUR DIFFERENT pipes: the next batch's recurrence queued on an audio
        stream's pipe waits for the oscillator kernel to finish dispatching (0.399 ms per step -> 0.47 / 0.53 with the second
        control stream on the first / second audio stream's pipe; period 4 in the number of queues created in between);
      * a fifth queue (the exchange of the multi-GPU path) shares a pipe with one of the four whatever happens: beside an audio
        stream three small launches per step on it cost 8-13 % of the step in every session; beside a control stream and created
        BEFORE it they were free in every session (x1.00-1.02), created after it free in one session and +11 % in another.
    Hence the first-use order  exchange, audio 0, audio 1, control 0, control 1  (consecutive queues: the four on four pipes,
    the exchange queue ahead of control 1 on its pipe), and ONE set per process: every ForwardPipeline of the same shape runs
    on the same streams (a second set would land on whatever pipes the creation count has reached - bench legs that built
    their own pipelines used to run up to 15 % slower than the same code alone)."""
    key = (torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device(), audio_streams,
           control_streams)
    got = _STREAM_SETS.get(key)
    if got is not None:
        return got
    xs = torch.cuda.Stream(device=dev)
    # a side stream only pays off if its work is small next to the audio half: high priority keeps its few workgroups
    # from queueing behind thousands of oscillator workgroups at dispatch
    audio = [torch.cuda.Stream(device=dev) for _ in range(audio_streams)]
    control = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(control_streams)]
    touch = torch.zeros(64, device=dev)
    order = [xs] + audio + control
    probe = os.environ.get("NWS_STREAM_ORDER")     # measurements (tools/queue_order_probe.sh): another first-use order, e.g.
    if probe:                                      # "a0,a1,c0,d,d,c1,x" - x exchange, aK / cK audio / control, d / h a dummy normal /
        order, dummies = [], []                    # high-priority stream that is never used again; unnamed streams follow in the default order
        for tok in probe.split(","):
            if tok in ("d", "h"):
                dummies.append(torch.cuda.Stream(device=dev, priority=-1 if tok == "h" else 0))
                order.append(dummies[-1])
            elif tok == "x":
                order.append(xs)
            elif tok[:1] in ("a", "c") and tok[1:].isdigit():
                lst = audio if tok[0] == "a" else control
                if int(tok[1:]) < len(lst):
                    order.append(lst[int(tok[1:])])
        order += [st for st in [xs] + audio + control if not any(st is o for o in order)]
    for st in order:                           # first use, in this order, one at a time
        with torch.cuda.stream(st):
            touch.fill_(0.0)
        st.synchronize()
    _STREAM_SETS[key] = (xs, audio, control)
    return _STREAM_SETS[key]


class ForwardPipeline:
    def __init__(self, model, depth: int = 4, audio_streams: int = 2, control_streams: int = 1, batched_gru: bool = False,
                 chain_exciters: bool = False):
        if depth < 2 or audio_streams < 1 or control_streams < 1:
            raise ValueError("need depth >= 2 and at least one stream of each kind")
        self.model = model
        self.eng = model._engine
        _, _, dev = self.eng.weights()
        self.dev = dev
        # streams placed on the command processor's pipes (placed_streams); `exchange` is for a multi-GPU caller's
        # parallel.CompletionDrivenExchange (unused otherwise: an idle queue costs nothing)
        self.exchange, self.audio, self.control = placed_streams(dev, audio_streams, control_streams)
        self.slots = [_Slot() for _ in range(depth)]
        self.batched_gru = batched_gru
        # The oscillator + waveshaper kernel saturates vector issue on every CU: two of them side by side (the audio halves
        # of neighbouring batches on two streams) only stretch each other, while the matrix / memory kernels of the OTHER
        # batch (frame MLPs before it, noise and reverb after it) fill the gaps it leaves.  With chain_exciters the oscillator
        # kernel of batch i+1 waits (stream-side, nws_forward_audio_ev) for the one of batch i.  Measured: no difference
        # (0.3919 vs 0.3932 ms/step) - the chip is work-bound, co-running kernels just share it - hence off by default.
        self.chain_exciters = bool(chain_exciters) and audio_streams > 1
        self._last_exciter = None
        self._n = 0
        self._shape = None
        self._outstanding = []

    def _prepare(self, B, T):
        if self._shape != (B, T):
            self.synchronize()
            for s in self.slots:
                s.ws = self.eng.new_workspace(B, T)
                s.used = False
                # draw buffers owned by the slot: no allocation and no cross-stream record_stream per step (after a device
                # synchronise the caching allocator walked hundreds of deferred-free events on the next allocation: 0.6 ms of
                # host time in front of the first step of a timed region)
                s.pu = torch.empty(_lib.N_HARMONICS, dtype=torch.float32, device=self.dev)
                s.nz = torch.empty(_lib.HOP * T - 1, dtype=torch.float32, device=self.dev)
            self._shape = (B, T)

    def next_audio_stream(self):
        """The stream the NEXT submit() will run its audio half on (work a caller wants ordered after that batch's reverb -
        an all-gather of its waveforms, say - is enqueued there)."""
        return self.audio[self._n % len(self.audio)]

    @staticmethod
    def row_blocks(B: int, chunks: int):
        """B rows in `chunks` blocks of even size >= 4 (the reverb transforms two utterances at a time); None if that does not
        divide"""
        if chunks <= 1 or B % chunks or (B // chunks) % 2 or B // chunks < 4:
            return None
        n = B // chunks
        return [(q * n, n) for q in range(chunks)]

    def submit(self, f0, control, *, phase_u=None, noise=None, out=None, row_blocks=None, on_block=None, generator=None,
               block_events=None):
        m = self.model
        f0 = _req(f0 if f0.is_contiguous() else f0.contiguous(), "f0")
        control = _req(control if control.is_contiguous() else control.contiguous(), "control")
        if f0.dim() != 3 or f0.shape[1] != 1 or control.dim() != 3 or control.shape[1] < 2:
            raise RuntimeError(f"expected f0 (B,1,T) and control (B,C>=2,T), got {tuple(f0.shape)} / {tuple(control.shape)}")
        B, _, T = f0.shape
        if control.shape[0] != B or control.shape[2] != T or T < 2:
            raise RuntimeError("f0 and control disagree on batch / frames (or fewer than 2 frames)")
        self._prepare(B, T)
        i = self._n
        self._n += 1
        slot = self.slots[i % len(self.slots)]
        cs = self.control[i % len(self.control)]
        au = self.audio[i % len(self.audio)]
        batched = bool(self.batched_gru)
        ready = torch.cuda.current_stream(self.dev).record_event()
        with torch.cuda.stream(cs):
            cs.wait_event(ready)
            if slot.used:
                cs.wait_event(slot.ev_audio)        # the previous tenant of this workspace has been consumed
            # the two hidden draws of forward(), in the reference's order, on the side stream: two more small launches that
            # the audio streams do not have to carry (the generator advances in submit order either way)
            # (same generator consumption as torch.rand_like / torch.rand: the same uniform_ kernel on 101 resp. 128 T - 1 elements)
            # (`generator`: the draws come from it instead of the device's default generator - ranks of a sharded batch pass
            # identically seeded generators and so draw the same values with no collective, parallel.make_shared_generator)
            pu = torch.rand(_lib.N_HARMONICS, out=slot.pu, generator=generator) if phase_u is None else phase_u      # RNG draw #1 (generators.py:55)
            pu = _req(pu.reshape(-1), "phase_u", _lib.N_HARMONICS)
            nz = torch.rand(m.control_hop * T - 1, out=slot.nz, generator=generator) if noise is None else noise     # RNG draw #2 (:30)
            nz = _req(nz, "noise", m.control_hop * T - 1)
            self.eng.forward_control(f0, control, slot.ws, batched_gru=batched)
            slot.ev_control.record(cs)
            # (the draws live in the slot's own buffers: the next tenant's control half waits for this batch's audio half
            # - slot.ev_audio above - before it overwrites them)
        with torch.cuda.stream(au):
            au.wait_event(ready)
            au.wait_event(slot.ev_control)
            if self.chain_exciters:
                out = self.eng.forward_audio(f0, B, T, pu, nz, slot.ws, out=out, wait_event=self._last_exciter,
                                             record_event=slot.ev_exciter)
                self._last_exciter = slot.ev_exciter
            else:
                out = self.eng.forward_audio(f0, B, T, pu, nz, slot.ws, out=out, row_blocks=row_blocks, on_block=on_block,
                                             block_events=block_events)
            slot.ev_audio.record(au)
        slot.used = True
        slot.keep = (f0, control, pu, nz)        # inputs stay alive until the slot is reused
        # outputs are only remembered for join_current_stream(): one ring's worth, so that the caching allocator sees a
        # steady set of blocks after depth + 1 submissions instead of a sawtooth of live 4 * B * T * 128-byte waveforms
        self._outstanding.append((out, slot.ev_audio))
        if len(self._outstanding) > len(self.slots):
            del self._outstanding[:-len(self.slots)]
        return out

    def join_current_stream(self):
        """Make the caller's current stream wait (asynchronously) for everything submitted so far."""
        cur = torch.cuda.current_stream(self.dev)
        for out, ev in self._outstanding[-len(self.slots):]:
            cur.wait_event(ev)
            out.record_stream(cur)
        self._outstanding.clear()

    def synchronize(self):
        for s in self.audio + self.control:
            s.synchronize()
        self._outstanding.clear()
