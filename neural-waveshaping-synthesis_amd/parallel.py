"""Multi-GPU rendering: a batch of independent utterances sharded over ranks (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no collective code; forward() is independent across the batch
dimension except that the two RNG draws are shared by all rows (generators.py:30,55), so
  * rank r renders rows [r*B/W, (r+1)*B/W) with the SAME phase_u / noise on every rank
    (drawn once on rank 0 and broadcast: 101 + N-1 floats), and
  * the rendered waveforms are all-gathered (the one exchange step BASELINE.json's north_star names).
`render_fn(f0, control, phase_u, noise) -> (b, N)` is the single-device forward; it is a parameter so the
CPU tests can drive this file with the oracle instead of the HIP engine.
"""
from __future__ import annotations

import os
import queue
import threading

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world: int, rank: int):
    """Contiguous, balanced split: the first (batch % world) ranks get one extra row."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def make_shared_generator(device, seed: int = 0x5EED):
    """A generator seeded identically on every rank: draws from it agree across ranks with no communication
    (same Philox seed and offset sequence), which keeps the per-step broadcast off the critical path."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def shared_draws(n_harmonics: int, n_noise: int, device, group=None, src: int = 0, generator=None):
    """The two hidden draws of forward(), identical on every rank: either drawn from an identically seeded
    `generator` on every rank (no collective), or drawn on `src` and broadcast."""
    if generator is not None:
        return (torch.rand(n_harmonics, device=device, generator=generator),      # draw #1 (generators.py:55)
                torch.rand(n_noise, device=device, generator=generator))          # draw #2 (generators.py:30)
    buf = torch.empty(n_harmonics + n_noise, dtype=torch.float32, device=device)
    if not dist.is_initialized() or dist.get_rank(group) == src:
        buf[:n_harmonics] = torch.rand(n_harmonics, device=device)   # draw #1 (generators.py:55)
        buf[n_harmonics:] = torch.rand(n_noise, device=device)       # draw #2 (generators.py:30)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(buf, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return buf[:n_harmonics], buf[n_harmonics:]


def render_sharded(render_fn, f0, control, group=None, phase_u=None, noise=None, gather: bool = True,
                   out: torch.Tensor | None = None, async_op: bool = False, force_collective: bool = False):
    """f0 (B,1,T), control (B,C,T) are the FULL batch on every rank (or pre-sharded with gather-only use).

    Returns the full (B, N) result on every rank (gather=True) or the local shard.  With
    ``async_op=True`` returns (out, work) so the all-gather of this step overlaps the next render.
    Equal shards use one all_gather_into_tensor; ragged shards fall back to all_gather of padded rows.
    ``force_collective=True`` issues the all-gather even at world size 1 (exercises the RCCL call on a 1-GPU box).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, _, T = f0.shape
    lo, hi = shard_bounds(B, world, rank)
    if phase_u is None or noise is None:
        raise ValueError("render_sharded needs the shared draws (see shared_draws())")
    local = render_fn(f0[lo:hi].contiguous(), control[lo:hi].contiguous(), phase_u, noise)
    if not gather or (world == 1 and not (force_collective and dist.is_initialized())):
        return (local, None) if async_op else local
    N = local.shape[-1]
    if B % world == 0:
        if out is None:
            out = torch.empty((B, N), dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
        return (out, work) if async_op else out
    rows = -(-B // world)
    pad = torch.zeros((rows, N), dtype=local.dtype, device=local.device)
    pad[: hi - lo] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    full = torch.cat([parts[r][: shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0]] for r in range(world)], 0)
    return (full, None) if async_op else full


def gather_full(full: torch.Tensor, y: torch.Tensor, group=None, async_op: bool = True):
    """Weak-scaling exchange of one step (bench.py --gpus N): every rank contributes its own (b, N) batch `y`; `full`
    (world * b, N) receives rank r's rows at [r * b, (r + 1) * b).  One all_gather_into_tensor on the current stream."""
    world = dist.get_world_size(group)
    if full.shape != (world * y.shape[0], y.shape[1]):
        raise ValueError(f"gather buffer {tuple(full.shape)} does not hold {world} shards of {tuple(y.shape)}")
    return dist.all_gather_into_tensor(full, y, group=group, async_op=async_op)


def gather_row_block(full: torch.Tensor, y: torch.Tensor, row0: int, n: int, group=None, async_op: bool = True):
    """Sub-batch exchange (`--gather-chunks`): rows [row0, row0 + n) of every rank's batch, as soon as their reverb
    is enqueued; rank r's block lands at rows [r * b + row0, r * b + row0 + n) of `full` - the same layout gather_full fills."""
    world = dist.get_world_size(group)
    b = y.shape[0]
    if full.shape != (world * b, y.shape[1]) or row0 < 0 or n <= 0 or row0 + n > b:
        raise ValueError(f"rows [{row0}, {row0 + n}) of {tuple(y.shape)} into {tuple(full.shape)}")
    views = [full[r * b + row0:r * b + row0 + n] for r in range(world)]
    return dist.all_gather(views, y[row0:row0 + n], group=group, async_op=async_op)


def gather_block_major(full: torch.Tensor, y: torch.Tensor, q: int, chunks: int, group=None, async_op: bool = True):
    """Sub-batch exchange as ONE all_gather_into_tensor per block: `full` (world * b, N) is read as (chunks, world, b / chunks, N)
    - block-major - so that what block q of every rank fills is contiguous (the list-of-views form of gather_row_block costs
    torch.distributed a flatten / unflatten copy pair per rank and block: a dozen small launches per step on the exchange
    queue, measured x1.64 on the step at world size 1 against x1.0x for this form).  `block_major_view` gives the consumer's
    (world, chunks, rows, N) view of the same memory."""
    world = dist.get_world_size(group)
    b = y.shape[0]
    if b % chunks or full.shape != (world * b, y.shape[1]) or not 0 <= q < chunks:
        raise ValueError(f"block {q} of {chunks} of {tuple(y.shape)} into {tuple(full.shape)}")
    n = b // chunks
    dst = full.view(chunks, world * n, y.shape[1])[q]
    return dist.all_gather_into_tensor(dst, y[q * n:(q + 1) * n], group=group, async_op=async_op)


def block_major_view(full: torch.Tensor, world: int, chunks: int) -> torch.Tensor:
    """(world, chunks, b / chunks, N) view of a gather buffer filled by gather_block_major: [r, q, j] = row q * (b / chunks) + j of
    rank r's batch"""
    rows, n_samples = full.shape
    n = rows // (world * chunks)
    return full.view(chunks, world, n, n_samples).permute(1, 0, 2, 3)


class _Ticket:
    __slots__ = ("issued", "done", "exc", "keep")

    def __init__(self):
        self.issued = threading.Event()     # the worker has enqueued the exchange (or failed)
        self.done = None                    # torch.cuda.Event behind the exchange on the exchange stream
        self.exc = None
        self.keep = None


class CompletionDrivenExchange:
    """Issues every batch's exchange from a helper thread, on a stream of its own, at the moment the HOST has seen the batch
    complete - so that no hardware queue ever sits on a device-side wait for another queue.

    Why (measured on MI355X, round 5, `tools/world1_diag.sh`): the obvious form - enqueue the all-gather right behind the
    batch's reverb, ordered by an event (`stream_x.wait_event(ev_audio)`; what torch.distributed does internally between the
    caller's stream and its NCCL stream) - leaves the exchange queue with an UNSATISFIED cross-queue barrier at its head for
    the whole step, every step.  With nothing to send (world size 1, even an empty `record -> wait -> record` hop in place of
    the collective) that alone cost the pipelined step +30 % (0.392 -> 0.518 ms) beside two control streams and two audio
    streams: the command processor's dispatch from the OTHER queues slows down while a queue is parked on a barrier packet.
    The same hop on an event the host has already seen complete is free (0.394 ms), and so is this class.

    Protocol (one instance per process / device):
      * `acquire(slot, stream)` before rows of gather buffer `slot` are rendered again: the host waits until the slot's
        previous exchange has been ISSUED (which implies its batch was complete), and `stream` waits for its completion on
        the device - by then usually long over, i.e. a satisfied wait;
      * render into the slot, record an event behind the batch, `post(slot, event, issue)`: the worker thread blocks on the
        event (host side, GIL released), then calls `issue()` with the exchange stream current.  `issue` enqueues the
        collective with `async_op=False` (torch.distributed then launches on the CURRENT stream: no internal stream hop) or
        the peer copies;
      * `drain()` at the end of a region.
    The submitting thread can run at most `nslots` batches ahead of the GPU (it has to: a gather buffer is only free once its
    exchange is out), which bounds the run-ahead exactly like the ring of workspaces of ForwardPipeline does.

    Lifetime of what is sent: `issue()` RETURNS the tensors its exchange reads (the rendered rows); the ticket holds them until
    `acquire()` of the same slot has made the rendering stream wait for the exchange's completion, or `drain()`.  Nothing else
    keeps them alive - synchronous collectives of torch >= 2.8 and PeerCopyAllGather's host-ordered pushes run on the exchange /
    copy streams without record_stream - so a tensor allocated from another stream's pool (a batch rendered out of place) would
    otherwise be handed to a later batch by the caching allocator while its exchange is still in flight.

    Failure: the FIRST exception of an `issue()` (or of waiting for its batch) poisons the worker.  Every later ticket completes
    with that exception WITHOUT calling its `issue()` - this rank must not skip one collective and then issue the following
    ones, which its peers would pair with their previous ones (same shapes: no error, silently wrong gather buffers) - `post()`
    raises at once on the submitting thread, and so do `acquire()` / `drain()`.  The process is expected to exit on that error
    (under torch.distributed.run the peers are then torn down instead of waiting in their next collective).
    Works without a GPU too (`device` cpu: no events, `issue` runs as soon as the worker gets to it) - the gloo tests."""

    def __init__(self, device, nslots: int, stream=None):
        """`stream`: the exchange stream (ForwardPipeline.exchange: placed on the command processor's pipes where its launches
        do not delay the pipeline's own, pipeline.placed_streams); default a new one."""
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.stream = (stream if stream is not None else torch.cuda.Stream(device=self.device)) if self.cuda else None
        self._tickets = [None] * int(nslots)
        self._failed = None                 # the first exception of the worker: nothing is issued after it (class docstring)
        self.profile = None                 # a list: the worker appends (wait, issue, record) seconds per exchange (diagnosis)
        self._all = []
        self._prune_at = 64
        self._q = queue.SimpleQueue()
        self._thread = threading.Thread(target=self._run, name="nws-exchange", daemon=True)
        self._thread.start()

    def _run(self):
        import time
        if self.cuda:
            torch.cuda.set_device(self.device)
            torch.cuda.set_stream(self.stream)          # thread-local: everything this thread enqueues goes to the exchange stream
        torch.set_grad_enabled(False)
        while True:
            item = self._q.get()
            if item is None:
                return
            t, ready, issue = item
            if self._failed is not None:                # poisoned: complete the ticket with the first failure, issue nothing
                t.exc = self._failed
                t.issued.set()
                continue
            try:
                t0 = time.perf_counter()
                if ready is not None:
                    ready.synchronize()                 # host-side: the rows are final (GIL released while waiting)
                t1 = time.perf_counter()
                t.keep = issue()
                t2 = time.perf_counter()
                if self.cuda:
                    t.done = self.stream.record_event()
                if self.profile is not None:
                    self.profile.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
            except BaseException as e:                  # handed to the submitting thread by post() / acquire() / drain()
                t.exc = e
                self._failed = e
            t.issued.set()

    def post(self, slot: int, ready, issue):
        """`ready`: torch.cuda.Event recorded behind the kernels that write the rows (None: nothing to wait for);
        `issue()`: enqueues the exchange on the current stream (called on the worker thread) and returns what it reads (kept
        alive by the ticket).  Raises at once when an earlier exchange has failed."""
        if self._failed is not None:
            raise RuntimeError("exchange worker failed earlier: no further exchange is issued") from self._failed
        t = _Ticket()
        self._tickets[slot] = t
        self._all.append(t)
        if len(self._all) >= self._prune_at:         # forget issued tickets now and then (amortised: the bound doubles with the backlog)
            self._all = [x for x in self._all if not x.issued.is_set() or x.exc is not None]
            self._prune_at = max(64, 2 * len(self._all))
        self._q.put((t, ready, issue))
        return t

    TIMEOUT_S = 600.0       # a batch or an exchange that has not completed by then never will: fail loudly instead of hanging

    @classmethod
    def _check(cls, t):
        if not t.issued.wait(timeout=cls.TIMEOUT_S):
            raise RuntimeError(f"exchange worker: an exchange was not issued within {cls.TIMEOUT_S:.0f} s (its batch never completed, "
                               f"or a collective is stuck waiting for a peer)")
        if t.exc is not None:
            raise RuntimeError("exchange worker failed") from t.exc

    def acquire(self, slot: int, stream=None):
        t = self._tickets[slot]
        if t is None:
            return
        self._check(t)
        if self.cuda and t.done is not None:
            (stream or torch.cuda.current_stream(self.device)).wait_event(t.done)
        t.keep = None                        # whatever renders into the slot next is ordered behind the exchange now
        self._tickets[slot] = None

    def drain(self, finalize=None):
        """Every posted exchange issued AND complete (host-side).  `finalize()`: run on the worker thread behind everything posted so
        far, before the wait (PeerCopyAllGather.flush of the host-ordered peer-copy form: the last exchange's completion signal)."""
        if finalize is not None:
            if self._failed is not None:
                raise RuntimeError("exchange worker failed earlier: no further exchange is issued") from self._failed
            t = _Ticket()
            self._all.append(t)
            self._q.put((t, None, finalize))
        for t in self._all:
            self._check(t)
        if self.cuda:
            self.stream.synchronize()
        for t in self._all:
            t.keep = None
        self._all = []
        self._prune_at = 64

    def close(self):
        self._q.put(None)
        self._thread.join(timeout=10)


class PeerCopyAllGather:
    """All-gather of the rendered waveforms WITHOUT collective kernels: every rank pushes its (b, N) shard straight into
    every peer's gather buffer with device-to-device copies (hipMemcpyAsync on peer-mapped memory = the SDMA copy
    engines over the point-to-point xGMI links, one link per peer: 7 x 16.4 MB in parallel at 64 clips per rank), so that
    no compute unit is taken from the oscillator kernels the way RCCL's ring/tree kernels take them.

    Set-up (once): each rank allocates `nbuf` gather buffers of (world * b, N), exports them as IPC handles
    (dmabuf IPC; HSA_ENABLE_IPC_MODE_LEGACY=0 must be set, as it is on the GPU boxes) and opens every peer's buffers.
    Per step: `gather(y, slot)` forks one copy per peer onto that peer's own copy stream (in-order streams would run the
    world - 1 copies one after the other: one link busy at a time), joins them back into the CURRENT stream, skips the local
    copy when the shard was rendered in place (`local_rows(slot)`), then one tiny stream-ordered collective
    (4-byte all-reduce; on the "gloo" test backend a host barrier after a stream sync) whose completion on rank q implies
    that every rank's copies into q's buffer were complete when that rank joined it.  Returns (full_buffer, work).

    Slot reuse is guarded here, not left to the caller: `gather` / `finish` hand slot s out, the consumer calls `release(s)`
    when the work it enqueued on full[s] is done with it (an event on its current stream), and
      * a new gather into a slot that was handed out and not released raises (host-side bookkeeping);
      * every rank makes its completion signal wait for its own release events, and every gather first waits for the previous
        gather's completion signal - so a push into rank q's slot s is ordered after q's last read of it (nbuf >= 2).
    """

    def __init__(self, rows: int, n_samples: int, device, nbuf: int = 2, group=None, dtype=torch.float32,
                 sync_signal: bool = False, fake_peers: int = 0, copy_streams=None, fake_rows: int = 0):
        """fake_peers (rehearsals on fewer GPUs than the job will have, `NWS_BENCH_FAKE_PEERS` of bench.py): that many extra
        destinations per push, each a LOCAL buffer on a copy stream of its own - the stream count and issue pattern of a world of
        `world + fake_peers` ranks, with this device's blit kernels standing in for the copy engines and links of real peers.
        copy_streams: one stream per destination (world + fake_peers) instead of fresh ones (a caller that places them).
        sync_signal (= host-ordered mode, for callers that issue every exchange from a helper thread on one stream of their
        own: CompletionDrivenExchange): nothing in an exchange is ordered by a device-side wait on another queue - such a wait,
        parked on a hardware queue, slows the dispatch of the queues next to it (LABBOOK round 5: +24-30 % on the pipelined step).
        The pushes go straight onto the per-peer copy streams (the rows are final: the caller has seen their batch complete), the
        issuing thread then waits ON THE HOST for its copies and for its own release events, and the completion signal is a
        synchronous collective = launched on the CURRENT stream (torch.distributed launches `async_op=False` collectives there:
        no hop to its NCCL stream and back, no work handle).
        The exchanges of successive steps are PIPELINED (round 6): `gather(y, slot)` first completes step i - 1 - a host-side wait
        on the copy streams (its pushes are a step old by now: satisfied waits, and no event record per copy), then its completion
        signal - and then pushes step i's rows (ONE nws_peer_push call: n copies without the interpreter lock) and returns: the
        copy engines work on step i while the pipeline renders step i + 1, and the helper thread never sits out a transfer.  (Round 5's form waited for every step's copies before it
        signalled and before the next step's copies could start: with seven destinations the helper thread needed 0.44-0.54 ms
        per 0.40 ms step and became the bottleneck, profiles/r06/fake_peers_ab.txt.)  Consequences for a consumer: full[slot] is
        complete on every rank once `complete(slot)` (or `flush()`) has returned and the issuing stream has been waited for; a
        slot must be released within nbuf - 2 steps."""
        self.sync_signal = bool(sync_signal)
        self._pending_copies = []            # host-ordered mode: events behind the pushes of the exchange being assembled
        self._deferred = []                  # host-ordered mode: (slot, copy events) of exchanges pushed but not yet signalled
        self._signalled = False              # host-ordered mode: a completion signal is in flight on the issuing stream
        if not dist.is_initialized():
            raise RuntimeError("PeerCopyAllGather needs an initialised process group (handle exchange)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.rows, self.n = int(rows), int(n_samples)
        self.device = torch.device(device)
        self.backend = dist.get_backend(group)
        self.full = [torch.empty((self.world * self.rows, self.n), dtype=dtype, device=self.device) for _ in range(nbuf)]
        # export / import through torch.multiprocessing's own CUDA-tensor sharing (the documented route by which a producer process
        # hands device tensors to a consumer: reduce_tensor -> (rebuild function, picklable IPC descriptor); dmabuf IPC handles
        # underneath).  The producer keeps `self.full` alive for the lifetime of this object, as that protocol requires.
        from torch.multiprocessing.reductions import reduce_tensor

        mine = [reduce_tensor(t) for t in self.full]
        everybody = [None] * self.world
        dist.all_gather_object(everybody, mine, group=group)
        self.remote = []        # remote[p][slot] = rank p's gather buffer, opened in this process
        for p, handles in enumerate(everybody):
            if p == self.rank:
                self.remote.append(self.full)
                continue
            opened = []
            for rebuild, args in handles:
                t = rebuild(*args)
                if tuple(t.shape) != (self.world * self.rows, self.n) or t.dtype != dtype:
                    raise RuntimeError(f"rank {p} exported a gather buffer of {tuple(t.shape)} {t.dtype}")
                opened.append(t)
            self.remote.append(opened)
        self.fake_peers = int(fake_peers)
        self.fake_rows = int(fake_rows)      # diagnosis: pushes carry this many rows only (the issue pattern without the data volume)
        for _ in range(self.fake_peers):     # destinations that stand in for peers this job does not have
            self.remote.append([torch.empty((self.world * self.rows, self.n), dtype=dtype, device=self.device) for _ in range(nbuf)])
        self.npush = self.world + self.fake_peers
        self._flag = torch.zeros(1, dtype=torch.float32, device=self.device)
        # one copy stream per peer: the pushes of a step run side by side, each on its own point-to-point link
        if copy_streams is not None and len(copy_streams) != self.npush:
            raise ValueError(f"copy_streams: one per destination ({self.npush}), got {len(copy_streams)}")
        self._copy_streams = (list(copy_streams) if copy_streams is not None else
                              [torch.cuda.Stream(device=self.device) for _ in range(self.npush)]
                              if self.device.type == "cuda" and self.npush > 1 else None)
        self.prof = {} if os.environ.get("NWS_PEER_PROF") == "1" else None      # diagnosis: seconds per section of gather()
        self._use_events = os.environ.get("NWS_PEER_EVENTS") == "1"      # A/B: an event per copy instead of a wait on the copy streams
        self._events = {}                    # host-ordered mode: (slot, destination, first row) -> reusable torch.cuda.Event
        self._held = [False] * nbuf          # handed to the consumer, not yet released
        self._released = [None] * nbuf       # event of the consumer's last use of the slot
        self._last_work = None               # completion signal of the previous gather
        dist.barrier(group=group)   # nobody starts pushing before everybody has opened everything

    def _claim(self, slot: int):
        if self._held[slot]:
            raise RuntimeError(f"PeerCopyAllGather: slot {slot} is still held by its consumer - call release({slot}) when the work "
                               f"reading full[{slot}] has been enqueued (peers would overwrite rows that are being read)")
        if self._last_work is not None:      # the previous exchange is complete on every rank before new rows travel
            self._last_work.wait()
            self._last_work = None
        if self.sync_signal and self._signalled and self.device.type == "cuda":
            # host side: the latest completion signal - of the exchange before last, enqueued a step ago - has landed (a stream
            # synchronise, not one more event record per step on the exchange queue: x1.08 instead of x1.01 at world size 1)
            if self.prof is not None:
                import time
                t0 = time.perf_counter()
            torch.cuda.current_stream(self.device).synchronize()
            if self.prof is not None:
                self.prof["wait_signal"] = self.prof.get("wait_signal", 0.0) + time.perf_counter() - t0
            self._signalled = False
        if self.sync_signal and not self._use_events:
            # the previous exchange's pushes have had a step to land: wait for them on the copy streams themselves (no event record
            # per copy: every record is one more packet for the command processor), signal, and only then start this step's pushes
            self._complete_deferred(keep=0)

    def release(self, slot: int, used: bool = True):
        """The consumer is done with full[slot]: everything it enqueued on the current stream so far may still read it, anything
        later must not.  used=False: the consumer enqueued NOTHING that reads the slot (a benchmark loop that only times the exchange):
        no event is recorded - every record is one more packet on the issuing queue."""
        if self.device.type == "cuda" and used:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._released[slot] = ev
        self._held[slot] = False

    def _signal(self, slot: int):
        if self.sync_signal:                         # host-ordered: this exchange's signal goes out in front of the NEXT one's pushes
            self._held[slot] = True
            self._deferred.append((slot, self._pending_copies))
            self._pending_copies = []
            if self._use_events:
                self._complete_deferred(keep=1)
            return self.full[slot], None
        st = torch.cuda.current_stream(self.device)
        for s, ev in enumerate(self._released):      # this rank's reads of released slots precede its completion signal
            if ev is not None:
                st.wait_event(ev)
                self._released[s] = None
        self._held[slot] = True
        if self.backend == "gloo":              # CPU-side test backend: no stream-ordered collectives
            st.synchronize()
            dist.barrier(group=self.group)
            return self.full[slot], None
        work = dist.all_reduce(self._flag, group=self.group, async_op=True)   # ordered after the copies on this stream
        self._last_work = work
        return self.full[slot], work

    def _complete_deferred(self, keep: int):
        """host-ordered mode: the completion signals of all but the `keep` newest pushed exchanges, oldest first.  Per exchange:
        this rank's copies have landed (host-side wait), its released slots are no longer being read (host-side wait), then the
        4-byte synchronous all-reduce on the current (exchange) stream."""
        import time
        while len(self._deferred) > keep:
            slot, evs = self._deferred.pop(0)
            t0 = time.perf_counter()
            if evs:
                import ctypes as C
                from . import _lib
                if self._use_events:
                    handles = (C.c_void_p * len(evs))(*[int(ev.cuda_event) for ev in evs])
                    _lib.check(_lib.lib().nws_events_wait(len(evs), handles), "nws_events_wait")
                else:                               # `evs` holds the copy streams' handles: the pushes are the last thing on them
                    handles = (C.c_void_p * len(evs))(*sorted(set(evs)))
                    _lib.check(_lib.lib().nws_streams_wait(len(handles), handles), "nws_streams_wait")
            t1 = time.perf_counter()
            for s, ev in enumerate(self._released):
                if ev is not None:
                    ev.synchronize()
                    self._released[s] = None
            t2 = time.perf_counter()
            if self.prof is not None:
                self.prof["wait_copies"] = self.prof.get("wait_copies", 0.0) + t1 - t0
                self.prof["wait_released"] = self.prof.get("wait_released", 0.0) + t2 - t1
            if self.backend == "gloo":              # CPU-side test backend: a host barrier is the signal
                dist.barrier(group=self.group)
                continue
            dist.all_reduce(self._flag, group=self.group, async_op=False)
            if self.prof is not None:
                self.prof["all_reduce_call"] = self.prof.get("all_reduce_call", 0.0) + time.perf_counter() - t2
            self._signalled = True

    def complete(self, slot: int):
        """host-ordered mode: make sure the completion signal of the exchange into `slot` has been issued (everything up to and
        including it); the consumer then orders its reads behind the issuing stream."""
        for k, (s, _) in enumerate(self._deferred):
            if s == slot:
                self._complete_deferred(keep=len(self._deferred) - k - 1)
                return

    def flush(self):
        """host-ordered mode: every pushed exchange signalled, and the last signal landed (host side).  Call it on the thread that
        issues the exchanges, at the end of a region - every rank, like the exchanges themselves."""
        self._complete_deferred(keep=0)
        if self._signalled and self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
            self._signalled = False

    def local_rows(self, slot: int) -> torch.Tensor:
        """This rank's own rows of full[slot]: render into them (forward(..., out=...)) and `gather` has nothing to copy locally.
        The current stream is made to wait for the consumer's last released use of the slot first."""
        ev = self._released[slot]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        lo = self.rank * self.rows
        return self.full[slot][lo:lo + self.rows]

    def _push(self, src: torch.Tensor, slot: int, lo: int):
        """src -> rows [lo, lo + len(src)) of every rank's full[slot]; ordered after the current stream's work so far, and the
        current stream continues after all of the copies"""
        n = src.shape[0]
        cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if self.sync_signal and self._copy_streams is not None:
            # host-ordered: no fork, no join - the copies start at once on their own streams (ONE call through the binding:
            # nws_peer_push), _complete_deferred waits for them on the host a step later
            import ctypes as C
            from . import _lib
            dsts, sts, evs = [], [], []
            for k in range(self.npush):
                p = (self.rank + k) % self.npush
                dst = self.remote[p][slot][lo:lo + n]
                if p == self.rank and dst.data_ptr() == src.data_ptr():
                    continue
                dsts.append(dst.data_ptr())
                sts.append(self._copy_streams[p].cuda_stream)
                if self._use_events:
                    evs.append(self._copy_event(slot, p, lo))
            if dsts:
                if not src.is_contiguous():
                    raise ValueError("PeerCopyAllGather: the rows to push must be contiguous")
                m = len(dsts)
                nbytes = (min(self.fake_rows, n) if self.fake_rows else n) * src.shape[1] * src.element_size()
                with torch.cuda.device(self.device):
                    _lib.check(_lib.lib().nws_peer_push(m, (C.c_void_p * m)(*dsts), src.data_ptr(), nbytes,
                                                        (C.c_void_p * m)(*sts),
                                                        (C.c_void_p * m)(*[int(e.cuda_event) for e in evs]) if evs else None),
                               "nws_peer_push")
                self._pending_copies += evs if self._use_events else sts
            return
        fork = cur.record_event() if self._copy_streams is not None else None
        joins = []
        for k in range(self.npush):             # start with the right-hand neighbour: the ranks' pushes spread over the links
            p = (self.rank + k) % self.npush
            dst = self.remote[p][slot][lo:lo + n]
            if p == self.rank and dst.data_ptr() == src.data_ptr():
                continue                        # rendered in place
            if self._copy_streams is None:
                dst.copy_(src, non_blocking=True)
                continue
            st = self._copy_streams[p]
            st.wait_event(fork)
            with torch.cuda.stream(st):
                dst.copy_(src, non_blocking=True)
                joins.append(st.record_event())
        for ev in joins:
            cur.wait_event(ev)

    def _copy_event(self, slot: int, p: int, lo: int):
        """the event behind the push of rows starting at `lo` of `slot` to destination p (created once, reused: nbuf exchanges
        later its previous use has long been waited for)"""
        key = (slot, p, lo)
        ev = self._events.get(key)
        if ev is None:
            ev = torch.cuda.Event()
            with torch.cuda.stream(self._copy_streams[p]):
                ev.record()                  # a torch event has no handle before its first record
            self._events[key] = ev
        return ev

    def push_rows(self, y: torch.Tensor, slot: int, row0: int, nrows: int):
        """Sub-batch form: push rows [row0, row0 + nrows) of this rank's shard into every peer's buffer (ordered
        after the current stream); call `finish(slot)` after the last block."""
        if row0 == 0:
            self._claim(slot)
        self._push(y[row0:row0 + nrows], slot, self.rank * self.rows + row0)

    def finish(self, slot: int):
        """the completion signal of gather(), on its own (after push_rows of every block)"""
        return self._signal(slot)

    def gather(self, y: torch.Tensor, slot: int):
        if y.shape != (self.rows, self.n) or not y.is_contiguous():
            raise ValueError(f"expected a contiguous {(self.rows, self.n)} shard, got {tuple(y.shape)}")
        if self.prof is not None:
            import time
            t0 = time.perf_counter()
            self._claim(slot)
            t1 = time.perf_counter()
            self._push(y, slot, self.rank * self.rows)
            t2 = time.perf_counter()
            r = self._signal(slot)
            t3 = time.perf_counter()
            for k, v in (("claim", t1 - t0), ("push", t2 - t1), ("signal", t3 - t2), ("n", 1)):
                self.prof[k] = self.prof.get(k, 0.0) + v
            return r
        self._claim(slot)
        self._push(y, slot, self.rank * self.rows)
        return self._signal(slot)
