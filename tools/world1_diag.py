#!/usr/bin/env python
"""bench.py with diagnosis switches (they used to live inside bench.py's timed closures; VERDICT r5 hygiene (c)): the SAME main(),
step function and JSON line, with hooks that replace / instrument the exchange.  Arguments are bench.py's own.

    NWS_BENCH_DIAG=wprof,noexch NWS_BENCH_FORCE_DIST=1 python tools/world1_diag.py --gather rccl --steps 200 ...

NWS_BENCH_DIAG (comma separated):
    noexch   the mechanism alone (events, helper thread), nothing issued
    blit3    three tiny launches per step on the exchange stream in place of the collective
    queued   round 4's form: the exchange enqueued behind the batch by a device-side wait on a side queue (+24-30 %)
    wprof    the helper thread's per-exchange times (wait / issue / record) -> exchange_worker_us
    peerprof seconds per section of PeerCopyAllGather.gather -> peer_gather_us (sets NWS_PEER_PROF=1)
    cprofile cProfile of the enqueue loop of every timed region -> stderr
    pgonly   an initialised RCCL communicator beside the plain single-GPU pattern (what does its mere presence cost?)
NWS_SWITCH=<seconds>: the interpreter's thread switch interval.
"""
import contextlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIAG = set(filter(None, os.environ.get("NWS_BENCH_DIAG", "").split(",")))
if os.environ.get("NWS_PEER_PROF") == "1":
    DIAG.add("peerprof")
if "peerprof" in DIAG:
    os.environ["NWS_PEER_PROF"] = "1"

import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402


class DiagHooks(bench.Hooks):
    def before_model(self, dev, distributed):
        if os.environ.get("NWS_SWITCH"):
            sys.setswitchinterval(float(os.environ["NWS_SWITCH"]))
        if "pgonly" in DIAG and not distributed:
            import torch.distributed as dist
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29512"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()

    def after_setup(self, dev, xchg, peer):
        self.dev = dev
        self.fake_side = torch.cuda.Stream(device=dev) if "queued" in DIAG else None
        self.tiny = [torch.zeros(64, device=dev) for _ in range(3)]
        if xchg is not None and "wprof" in DIAG:
            xchg.profile = []

    def post_behind(self, xchg, slot_i, ev, issue):
        if "noexch" in DIAG:
            xchg.post(slot_i, ev, lambda: None)
            return True
        if "blit3" in DIAG:
            def blits():
                self.tiny[0].fill_(0.0)
                self.tiny[1].fill_(1.0)
                self.tiny[2].copy_(self.tiny[0])
            xchg.post(slot_i, ev, blits)
            return True
        if "queued" in DIAG:
            self.fake_side.wait_event(ev)
            with torch.cuda.stream(self.fake_side):
                issue()
            return True
        return False

    @contextlib.contextmanager
    def around_timed(self):
        if "cprofile" not in DIAG:
            yield
            return
        import cProfile
        import pstats
        prof = cProfile.Profile()
        prof.enable()
        try:
            yield
        finally:
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(45)

    def extra(self, extra, steps, xchg, peer):
        if xchg is not None and xchg.profile:
            pr = np.array(xchg.profile[-steps:]) * 1e6
            extra["exchange_worker_us"] = {"wait_p50": float(np.median(pr[:, 0])), "issue_p50": float(np.median(pr[:, 1])),
                                           "record_p50": float(np.median(pr[:, 2])), "issue_mean": float(pr[:, 1].mean())}
        if peer is not None and peer.prof:
            extra["peer_gather_us"] = {k: round(v / max(1.0, peer.prof.get("n", 1.0)) * 1e6, 1) for k, v in peer.prof.items() if k != "n"}


if __name__ == "__main__":
    bench.main(DiagHooks())
