"""The MI355X co-execution hazard (csrc/coexec_probe.hip, DESIGN.md 5.3, LABBOOK.md '5.2') and the product's immunity to it.

A v_pk_{add,mul,fma}_f32 whose low lane reads the high half of src1 returns wrong values while another kernel runs
K=16/32 f16 MFMAs on the same CU.  The build keeps that form out of every product kernel; these tests check
  * the probe itself (all eight forms agree with scalar arithmetic when nothing runs beside it; the four forms the
    product is allowed to use stay correct beside every MFMA flavour),
  * the path that was hit: successive forwards on TWO audio streams (reverb of batch i beside the frame-MLP / noise
    kernels of batch i+1) reproduce the plain forward bit for bit.
Whether forms 4..7 actually fail on the box at hand is reported, not asserted: that is the hardware's business.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))

pytestmark = pytest.mark.gpu


def test_probe_forms_and_safe_forms_beside_mfma():
    import coexec_probe
    r = coexec_probe.run(blocks=2048, iters=1000, rounds=3, product_kernels=False)
    wrong = r["wrong_results"]
    assert wrong["none"] == [0] * 8, wrong["none"]
    for load, counts in wrong.items():
        assert counts[:4] == [0, 0, 0, 0], (load, counts)      # the forms the build lets through
    hazard = {k: v[4:] for k, v in wrong.items() if any(v[4:])}
    print("swizzled-src1 forms wrong beside:", hazard or "nothing on this box")
    # DESIGN.md 5.3, LABBOOK.md '5.2' as falsifiable statements, with the whole matrix left in gpurun_out/parity_report.json either way:
    #   (i) the swizzled-src1 forms are exact beside the MFMAs that predate gfx950 (32x32x8 f16, 32x32x2 f32) and beside nothing;
    #  (ii) IF a swizzled form fails on this box at all, it fails beside a K=16 / K=32 half-precision MFMA (the claim is about
    #       those instructions: a failure anywhere else would be a different hazard the build guard does not describe).
    from gpu_util import record
    record("coexec_matrix", forms=r["forms"], evaluations_per_form=r["evaluations_per_form"], wrong_results=wrong,
           wrong_results_other_families=r["wrong_results_other_families"], hazard_reproduced_on_this_box=bool(hazard))
    for old_mfma in ("v_mfma_f32_32x32x8f16", "v_mfma_f32_32x32x2f32"):
        assert wrong[old_mfma][4:] == [0, 0, 0, 0], (old_mfma, wrong[old_mfma])
    new_mfma = ("v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x32_f16", "same kernel: MFMA in waves 2-3 (x0.5 evals)")
    assert all(k in new_mfma for k in hazard), hazard
    other = r["wrong_results_other_families"]
    assert other["none"] == [0] * 11, other["none"]          # the second probe agrees with itself
    for load, counts in other.items():
        assert counts[7] == 0, (load, counts)               # plain packed fp16 (control column)
    print("other families wrong beside:", {k: v for k, v in other.items() if any(v)} or "nothing on this box")


def test_two_audio_streams_bit_exact_soak():
    from gpu_util import build_model
    import nws_amd
    m = build_model(True)
    B, T = 64, 500
    g = torch.Generator(device="cuda").manual_seed(5)
    n = 6
    batches = []
    for _ in range(n):
        f0 = (100 + 900 * torch.rand(B, 1, 1, device="cuda", generator=g)) * (1 + 0.01 * torch.randn(B, 1, T, device="cuda", generator=g))
        c = torch.randn(B, 2, T, device="cuda", generator=g)
        pu = torch.rand(101, device="cuda", generator=g)
        nz = torch.rand(128 * T - 1, device="cuda", generator=g)
        batches.append((f0, c, pu, nz))
    eng = m._engine
    refs = []
    for f0, c, pu, nz in batches:
        ws = eng.new_workspace(B, T)
        eng.forward_control(f0, c, ws, batched_gru=False)
        refs.append(eng.forward_audio(f0, B, T, pu, nz, ws).clone())
    torch.cuda.synchronize()
    pipe = nws_amd.ForwardPipeline(m, depth=3, audio_streams=2)
    mismatching = 0
    rounds = 40
    for r in range(rounds):
        outs = [pipe.submit(f0, c, phase_u=pu, noise=nz) for f0, c, pu, nz in batches]
        pipe.synchronize()
        mismatching += sum(0 if torch.equal(o, ref) else 1 for o, ref in zip(outs, refs))
    assert mismatching == 0, f"{mismatching} of {rounds * n} batches differ from the plain forward"


def test_torch_rng_draws_unchanged_beside_mfma_load():
    """The two hidden draws of forward() come from torch's own kernels, which the build guard cannot inspect; in the pipeline
    they run on the control stream beside the audio half's MFMA kernels.  Same seed -> same bits, with and without a
    half-precision MFMA loop on the next stream."""
    import nws_amd
    _lib = nws_amd._lib
    L = _lib.probe_lib()
    s_draw, s_load = torch.cuda.Stream(), torch.cuda.Stream()
    sink = torch.zeros(256, device="cuda")
    gen = torch.Generator(device="cuda")

    def draws():
        gen.manual_seed(7)
        with torch.cuda.stream(s_draw):
            out = [(torch.rand(101, device="cuda", generator=gen), torch.rand(63999, device="cuda", generator=gen))
                   for _ in range(40)]
        return out

    ref = draws()
    torch.cuda.synchronize()
    for kind in (0, 1):
        for _ in range(3):
            _lib.check(L.nws_coexec_mfma_load(kind, 8192, 3000, sink.data_ptr(), s_load.cuda_stream), "load")
            got = draws()
            torch.cuda.synchronize()
            assert all(torch.equal(a, c) and torch.equal(b, d) for (a, b), (c, d) in zip(got, ref)), kind


def test_every_pipeline_of_a_process_runs_on_one_placed_stream_set():
    """pipeline.placed_streams: ONE measured stream set per process and device; every ForwardPipeline - whatever its stream counts -
    runs on prefixes of it (round 5 kept one set per SHAPE: a second shape sat on whatever pipes the creation count had reached,
    +18-35 % per step).  `streams=` opts out with private streams."""
    import importlib

    from gpu_util import build_model

    pm = importlib.import_module("neural-waveshaping-synthesis_amd.pipeline")
    m = build_model(True)
    p1 = pm.ForwardPipeline(m, depth=4, audio_streams=2, control_streams=2)
    p2 = pm.ForwardPipeline(m, depth=3, audio_streams=2, control_streams=2)
    assert p1.exchange is p2.exchange and all(a is b for a, b in zip(p1.audio + p1.control, p2.audio + p2.control))
    assert len({s.cuda_stream for s in [p1.exchange] + p1.audio + p1.control}) == 5          # five distinct HIP streams
    assert all(s.priority == -1 for s in p1.control) and all(s.priority == 0 for s in p1.audio + [p1.exchange])
    p3 = pm.ForwardPipeline(m, depth=4, audio_streams=2, control_streams=1)                  # another shape: a prefix of the same set
    assert p3.audio[0] is p1.audio[0] and p3.audio[1] is p1.audio[1] and p3.control == p1.control[:1]
    with pytest.raises(ValueError):
        pm.ForwardPipeline(m, audio_streams=3)
    mine = (torch.cuda.Stream(), [torch.cuda.Stream()], [torch.cuda.Stream(priority=-1)])
    p4 = pm.ForwardPipeline(m, depth=3, streams=mine)                                        # private streams: the caller's business
    assert p4.audio[0] is mine[1][0] and p4.control[0] is mine[2][0] and p4.audio[0] is not p1.audio[0]
    rep = pm.placement_report("cuda")
    # (queue_offset is informational: None when no normal candidate shares the submitting stream's pipe - streams the earlier tests of
    # this process created and dropped hand their hardware queues to later streams, so candidates need not arrive round-robin)
    assert rep["mode"] == "probe" and rep["ok"] and rep["verified"] and rep["queue_offset"] in (None, 0, 1, 2, 3), rep
    # two pipelines sharing the streams (and a private one) still return the plain forward's bits
    g = torch.Generator(device="cuda").manual_seed(8)
    f0 = 100 + 600 * torch.rand(4, 1, 40, device="cuda", generator=g)
    c = torch.randn(4, 2, 40, device="cuda", generator=g)
    pu, nz = torch.rand(101, device="cuda", generator=g), torch.rand(128 * 40 - 1, device="cuda", generator=g)
    with torch.no_grad():
        ref = m(f0, c, phase_u=pu, noise=nz)
        ys = [p.submit(f0, c, phase_u=pu, noise=nz) for p in (p1, p2, p1, p2, p3, p4)]
        for p in (p1, p2, p3, p4):
            p.synchronize()
    assert all(torch.equal(y, ref) for y in ys)
    assert pm.verify_placement("cuda")["ok"]               # the queues still sit where they were found


def _placement_case(*args, env=None):
    import json
    import subprocess
    e = dict(os.environ, **(env or {}))
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "placement_case.py"), *args],
                       capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_placement_does_not_depend_on_the_process_history():
    """VERDICT r5 #1a: the plain pipelined step (B = 64 x 4 s) after 0 / 1 / 2 / 3 streams used earlier in the process, after a
    pipeline of another shape, and after both - one process per history.  Every placement verifies, the queue offset found is the
    number of earlier queues (mod 4), and every step is within 3 % of the best.  For scale, one misplaced set (round 5's
    first-use-order switch with a stream first used between the control streams) must be visibly slower on the same box."""
    from gpu_util import record
    cases = {"pre0": [], "pre1": ["--pre", "1"], "pre2": ["--pre", "2"], "pre3": ["--pre", "3"], "shape_first": ["--shape-first"],
             "pre2_shape_first": ["--pre", "2", "--shape-first"]}
    got = {k: _placement_case(*v) for k, v in cases.items()}
    for k, r in got.items():
        p = r["placement"]
        assert p["mode"] == "probe" and p["ok"] and p["verified"] and r["recheck"], (k, r)
        # every used stream shifts the offset found by one (relative to the run without any: the hardware queues of OTHER
        # processes - this pytest process - count towards the pipes too)
        assert (p["queue_offset"] - got["pre0"]["placement"]["queue_offset"]) % 4 == r["pre"] % 4, (k, p, got["pre0"]["placement"])
    best = min(r["ms_per_step"] for r in got.values())
    for k in cases:                      # up to two more runs for an outlier before it fails the 3 % bar (a process is a 150-step region)
        for _ in range(2):
            if got[k]["ms_per_step"] <= 1.03 * best:
                break
            again = _placement_case(*cases[k])
            if again["ms_per_step"] < got[k]["ms_per_step"]:
                got[k] = again
    bad = _placement_case(env={"NWS_STREAM_ORDER": "x,a0,a1,c0,d,c1"})
    record("placement_histories", ms_per_step={k: r["ms_per_step"] for k, r in got.items()}, misplaced_ms_per_step=bad["ms_per_step"])
    for k, r in got.items():
        assert r["ms_per_step"] <= 1.03 * best, (k, r["ms_per_step"], best)
    assert bad["ms_per_step"] > 1.05 * best, (bad["ms_per_step"], best)      # measured +14 ... +33 %
