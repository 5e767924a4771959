"""-m gpu parity tests: every HIP stage and the whole forward against the oracle / golden vectors.

All calls go through the C-ABI (libnws_hip.so via ctypes).  Tolerances are written next to each check;
the end-to-end bar is BASELINE.json's: <= 1e-4 RMS against the reference CPU forward on identical
F0 / control / checkpoint / RNG draws (golden vectors recorded from the real reference).
"""
import math
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_npz, rms
from gpu_util import build_model, dev, maxabs, record

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle(weights):
    from oracle.newt_oracle import OracleNEWT

    return OracleNEWT(weights, fast=False), OracleNEWT(weights, fast=True, lut_python_loop=False)


@pytest.fixture(scope="module")
def models():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return build_model(False), build_model(True)


def test_library_loaded_and_mfma_layout():
    import nws_amd as nws
    from importlib import import_module

    lib = import_module("neural-waveshaping-synthesis_amd._lib")
    assert lib.lib().nws_abi_version() == lib.ABI_VERSION
    bad = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    lib.check(lib.lib().nws_selftest_mfma(bad.data_ptr(), lib.stream_ptr()))
    assert int(bad.item()) == 0


def test_sin_accuracy_all_ranges():
    from importlib import import_module

    lib = import_module("neural-waveshaping-synthesis_amd._lib")
    rng = np.random.default_rng(0)
    worst = {}
    for name, scale in (("1e1", 1e1), ("1e3", 1e3), ("1e5", 1e5), ("5e6", 5e6), ("1e8", 1e8)):
        x = (rng.uniform(-scale, scale, 1 << 18)).astype(np.float32)
        xd = dev(x)
        y = torch.empty_like(xd)
        lib.check(lib.lib().nws_sin(xd.data_ptr(), y.data_ptr(), xd.numel(), lib.stream_ptr()))
        err = maxabs(y.cpu().numpy(), np.sin(x.astype(np.float64)))
        worst[name] = err
        assert err <= 3e-7, (name, err)   # torch's CPU sin (Sleef u10) is within ~6e-8 of the same reference
    record("sin_max_abs_err", **worst)


def test_phase_carry_matches_double_cumsum(models, oracle):
    m, _ = models
    g = load_npz("g6_highf0.npz")
    st = {}
    oracle[0](g["f0"], g["control"], load_npz("g1_realistic.npz")["phase_u"], load_npz("g1_realistic.npz")["noise"], stages=st)
    f0_up = st["f0_up"].numpy().astype(np.float64)
    ref = np.concatenate([[0.0], np.cumsum(f0_up[0])])[:-1][::32]
    carry = m._engine.phase_carry(f0=dev(g["f0"][:, 0])).cpu().numpy()[0]
    assert carry.shape == ref.shape
    assert np.array_equal(carry, ref)          # fp64 sums of fp32 values: exact, any order
    # pre-upsampled F0 path (public render_exciter) gives the same carries
    carry2 = m._engine.phase_carry(f0_up=dev(st["f0_up"].numpy())).cpu().numpy()[0]
    assert np.array_equal(carry2, ref)


def test_fused_gru_and_carry_launch(models):
    """nws_control_gru_carry = nws_control_gru + nws_phase_carry in one launch: identical bits."""
    import ctypes as C
    import nws_amd
    _lib = nws_amd._lib
    m, _ = models
    eng = m._engine
    w, _, _ = eng.weights()
    g = torch.Generator().manual_seed(3)
    for B, T in ((1, 2), (3, 37), (5, 1100)):
        f0 = (100 + 900 * torch.rand(B, 1, T, generator=g)).cuda()
        control = torch.randn(B, 3, T, generator=g).cuda()
        gru = torch.empty(B, T, 128, device="cuda")
        carry = torch.empty(B, 4 * T, dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib().nws_control_gru_carry(C.byref(w), _lib.ptr(control), _lib.ptr(f0), B, 3, T, _lib.ptr(gru),
                                                    _lib.ptr(carry), _lib.stream_ptr()), "nws_control_gru_carry")
        assert torch.equal(gru, eng.control_gru(control))
        assert torch.equal(carry, eng.phase_carry(f0=f0[:, 0].contiguous()))


def test_exciter_stage(models, oracle):
    m, _ = models
    worst = {}
    for name in ("g3_stages.npz", "g1_realistic.npz", "g6_highf0.npz", "g2_rand.npz"):
        g = load_npz(name)
        d = g if "phase_u" in g else load_npz("g1_realistic.npz")
        st = {}
        oracle[0](g["f0"], g["control"], d["phase_u"], d["noise"], stages=st)
        f0 = dev(g["f0"][:, 0])
        carry = m._engine.phase_carry(f0=f0)
        exc, _ = m._engine.exciter_newt(f0, None, carry, dev(d["phase_u"]), None, want_exciter=True, want_newt=False)
        err = maxabs(exc.cpu().numpy(), st["exciter"].numpy())
        worst[name] = err
        # bit-exact argument chain + <=1.5e-7 sin error, 101-term fp32 contraction of O(1) weights
        assert err <= 2e-5, (name, err)
    record("exciter_max_abs_err", **worst)
    g = load_npz("g3_stages.npz")
    exc = m._engine.exciter_newt(dev(g["f0"][:, 0]), None, m._engine.phase_carry(f0=dev(g["f0"][:, 0])),
                                 dev(g["phase_u"]), None, want_exciter=True, want_newt=False)[0]
    assert maxabs(exc.cpu().numpy(), g["exciter"]) <= 2e-5   # against the reference's own tap


def _gru_float64(w, control):
    """float64 GRU (gate order r,z,n; h' = (h - n) z + n): the yard-stick both fp32 implementations are held to."""
    Wi, Wh = w["embedding.gru.weight_ih_l0"].astype(np.float64), w["embedding.gru.weight_hh_l0"].astype(np.float64)
    bi, bh = w["embedding.gru.bias_ih_l0"].astype(np.float64), w["embedding.gru.bias_hh_l0"].astype(np.float64)
    B, _, T = control.shape
    h = np.zeros((B, 128))
    out = np.zeros((B, T, 128))
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))  # noqa: E731
    for t in range(T):
        gi = control[:, :2, t].astype(np.float64) @ Wi.T + bi
        gh = h @ Wh.T + bh
        r, z = sig(gi[:, :128] + gh[:, :128]), sig(gi[:, 128:256] + gh[:, 128:256])
        n = np.tanh(gi[:, 256:] + r * gh[:, 256:])
        h = (h - n) * z + n
        out[:, t] = h
    return out


def test_gru_and_frame_mlps(models, oracle, weights):
    m, _ = models
    worst = {}
    for name in ("g3_stages.npz", "g1_realistic.npz", "g2_rand.npz"):
        g = load_npz(name)
        d = g if "phase_u" in g else load_npz("g1_realistic.npz")
        st = {}
        oracle[0](g["f0"], g["control"], d["phase_u"], d["noise"], stages=st)
        gru = m._engine.control_gru(dev(g["control"]))
        e_gru = maxabs(gru.cpu().numpy(), st["gru_out"].numpy())
        g64 = _gru_float64(weights, g["control"])
        e_gru64, e_ref64 = maxabs(gru.cpu().numpy(), g64), maxabs(st["gru_out"].numpy(), g64)
        # stage isolation: the MLP kernel is fed the oracle's GRU output, so its check is not polluted by the
        # (legitimate) fp32 divergence of two recurrent implementations
        emb, film, H, fir = m._engine.frame_mlps(st["gru_out"].contiguous().cuda(), want_emb=True, want_H=True)
        e_emb = maxabs(emb.cpu().numpy(), st["embedding"].numpy())
        e_film = maxabs(film.cpu().numpy(), st["film"].numpy().transpose(0, 2, 1))
        e_H = maxabs(H.cpu().numpy(), st["H"].numpy().transpose(0, 2, 1))
        Ht = st["H"].transpose(1, 2)
        h = torch.fft.irfft(torch.complex(Ht, torch.zeros_like(Ht))).roll(128, -1) * torch.hann_window(256).view(1, 1, -1)
        # the kernels hand over the upper half of the mirror-symmetric taps (include/nws_hip.h): the oracle's own full rows are
        # symmetric about tap 128 with h[0] = 0 to rounding, which is what makes that legitimate
        assert fir.shape[-1] == 128
        assert maxabs(h[..., 1:128].numpy(), h[..., 129:].flip(-1).numpy()) <= 1e-6 * max(1.0, float(h.abs().max())) and float(h[..., 0].abs().max()) == 0.0
        e_fir = maxabs(fir.cpu().numpy(), h[..., 128:].numpy())
        worst[name] = dict(gru=e_gru, gru_vs_f64=e_gru64, torch_gru_vs_f64=e_ref64, emb=e_emb, film=e_film, H=e_H, fir=e_fir)
        record("frame_path_max_abs_err", **worst)
        # 500 recurrent fp32 steps: torch's own CPU GRU sits 1-2e-5 (max-abs) from a float64 GRU on these inputs;
        # the HIP kernel must be in the same class (<= 3x torch's error, floor 5e-6)
        assert e_gru64 <= max(5e-6, 3.0 * e_ref64), worst
        assert e_emb <= 1e-5, worst                            # one 128-term fp32 contraction
        assert e_film <= 5e-5 and e_H <= 5e-5, worst          # 4 layers + LayerNorm
        assert e_fir <= 2e-6 * max(1.0, float(np.abs(st["H"].numpy()).max())), worst
    record("frame_path_max_abs_err", **worst)
    g = load_npz("g3_stages.npz")
    emb = m.get_embedding(dev(g["control"]))           # public API: GRU + proj, against the reference's own tap
    assert emb.shape == (2, 128, 3)
    assert maxabs(emb.cpu().numpy(), g["embedding"]) <= 1e-5


@pytest.mark.parametrize("mode", [1, 2])
def test_frame_mlp_kernels_against_the_stage_taps(models, oracle, mode):
    """Both frame-MLP kernel families forced on the same inputs (nws_debug_frame_mlps_kernel: 1 = one M-tile per wave with the
    activations in LDS, 2 = wave-resident frames, round 4): the reference's stage taps at T = 8 and T = 500 (two and 500 frames
    of a 256-frame workgroup: partial waves, empty waves), and a batch whose 65 x 501 frames end inside a wave."""
    from nws_amd import _lib
    m, _ = models
    L = _lib.lib()
    assert L.nws_debug_frame_mlps_kernel(mode) == 0
    try:
        for name in ("g3_stages.npz", "g1_realistic.npz"):
            g = load_npz(name)
            d = g if "phase_u" in g else load_npz("g1_realistic.npz")
            st = {}
            oracle[0](g["f0"], g["control"], d["phase_u"], d["noise"], stages=st)
            emb, film, H, fir = m._engine.frame_mlps(st["gru_out"].contiguous().cuda(), want_emb=True, want_H=True)
            film2, fir2 = m._engine.frame_mlps(st["gru_out"].contiguous().cuda())[1::2]      # the forward's instantiation (no taps)
            assert torch.equal(film, film2) and torch.equal(fir, fir2)
            Ht = st["H"].transpose(1, 2)
            h = torch.fft.irfft(torch.complex(Ht, torch.zeros_like(Ht))).roll(128, -1) * torch.hann_window(256).view(1, 1, -1)
            errs = dict(emb=maxabs(emb.cpu().numpy(), st["embedding"].numpy()),
                        film=maxabs(film.cpu().numpy(), st["film"].numpy().transpose(0, 2, 1)),
                        H=maxabs(H.cpu().numpy(), st["H"].numpy().transpose(0, 2, 1)),
                        fir=maxabs(fir.cpu().numpy(), h[..., 128:].numpy()))
            record(f"frame_mlps_mode{mode}_{name[:2]}", **errs)
            assert errs["emb"] <= 1e-5 and errs["film"] <= 5e-5 and errs["H"] <= 5e-5, errs
            assert errs["fir"] <= 2e-6 * max(1.0, float(np.abs(st["H"].numpy()).max())), errs
        gen = torch.Generator().manual_seed(7)
        gru = torch.tanh(torch.randn(65, 501, 128, generator=gen))
        _, film, _, fir = m._engine.frame_mlps(gru.cuda())
        L.nws_debug_frame_mlps_kernel(1)
        _, film_t, _, fir_t = m._engine.frame_mlps(gru.cuda())
        e_film, e_fir = maxabs(film.cpu().numpy(), film_t.cpu().numpy()), maxabs(fir.cpu().numpy(), fir_t.cpu().numpy())
        record(f"frame_mlps_mode{mode}_vs_tiles", film=e_film, fir=e_fir, film_max=float(film_t.abs().max()), fir_max=float(fir_t.abs().max()))
        assert e_film <= 2e-5 and e_fir <= 2e-6 * max(1.0, float(fir_t.abs().max()) * 50), (e_film, e_fir)
    finally:
        L.nws_debug_frame_mlps_kernel(0)


@pytest.mark.parametrize("B,T", [(1, 2), (3, 2), (5, 1), (17, 2)])
def test_frame_mlps_of_one_or_two_frames_against_the_tile_kernel(models, B, T):
    """Utterances of one or two frames (a streaming hop, a 256-sample buffer) take the matrix-vector form of csrc/mlp_few.h: one
    workgroup of four waves per (utterance, path), the same fp16 fragment pairs as the MFMA kernels (hi + lo = the weight to 22
    bits) against fp32 activations.  Forced back onto the 32-frame tile kernel (nws_debug_frame_mlps_kernel(1)) the same inputs
    must give the same FiLM rows and FIR half-taps to fp32 rounding; three frames take the tile kernel either way."""
    from nws_amd import _lib
    m, _ = models
    m._engine.weights()
    L = _lib.lib()
    gen = torch.Generator().manual_seed(100 * B + T)
    gru = torch.tanh(torch.randn(B, T, 128, generator=gen)).cuda()
    try:
        _, film, _, fir = m._engine.frame_mlps(gru)
        assert L.nws_debug_frame_mlps_kernel(1) == 0
        _, film_t, _, fir_t = m._engine.frame_mlps(gru)
    finally:
        L.nws_debug_frame_mlps_kernel(0)
    assert film.shape == (B, T, 256) and fir.shape == (B, T, 128) and torch.isfinite(film).all() and torch.isfinite(fir).all()
    e_film, e_fir = maxabs(film.cpu().numpy(), film_t.cpu().numpy()), maxabs(fir.cpu().numpy(), fir_t.cpu().numpy())
    record(f"frame_mlps_few_B{B}_T{T}", film=e_film, fir=e_fir, film_max=float(film_t.abs().max()), fir_max=float(fir_t.abs().max()))
    assert e_film <= 2e-5 * max(1.0, float(film_t.abs().max())) and e_fir <= 2e-6 * max(1.0, float(fir_t.abs().max()) * 50), (e_film, e_fir)
    assert not torch.equal(film, film_t) or B * T == 0          # it IS another kernel (same bits would mean the switch did nothing)


def test_forward_pipeline_matches_plain_forward(models, oracle):
    """ForwardPipeline (control half on side streams, batched GRU, ring of workspaces) must return what model() returns for
    the same inputs and draws, batch after batch, including a shape change in mid-stream; one batch is also held against
    the oracle."""
    import nws_amd
    _, fast = models
    g = torch.Generator().manual_seed(31)
    jobs = []
    for k, (B, T) in enumerate([(20, 40), (20, 40), (20, 40), (20, 40), (20, 40), (3, 17), (3, 17), (33, 24)]):
        f0 = (100 + 600 * torch.rand(B, 1, 1, generator=g)) * (1 + 0.01 * torch.randn(B, 1, T, generator=g))
        control = torch.randn(B, 2, T, generator=g)
        pu, nz = torch.rand(101, generator=g), torch.rand(128 * T - 1, generator=g)
        jobs.append((f0.cuda(), control.cuda(), pu.cuda(), nz.cuda()))
    torch.cuda.synchronize()
    refs = [fast(f0, c, phase_u=pu, noise=nz).cpu().numpy() for f0, c, pu, nz in jobs]
    f0, c, pu, nz = jobs[0]
    ref_orc = oracle[1](f0.cpu(), c.cpu(), pu.cpu(), nz.cpu()).numpy()
    for batched, kw in ((False, dict()),                                              # defaults: one audio, one control stream
                        (False, dict(depth=4, audio_streams=1, control_streams=2)),
                        (False, dict(depth=4, audio_streams=2, control_streams=1)),
                        (False, dict(depth=4, audio_streams=2, control_streams=2, chain_exciters=True)),   # nws_forward_audio_ev hooks
                        (True, dict(depth=2, audio_streams=1, control_streams=2))):
        pipe = nws_amd.ForwardPipeline(fast, batched_gru=batched, **kw)
        outs = [pipe.submit(f0, c, phase_u=pu, noise=nz) for f0, c, pu, nz in jobs]
        pipe.join_current_stream()
        got = [o.cpu().numpy() for o in outs]
        worst = max(rms(y - r) / max(rms(r), 1e-9) for y, r in zip(got, refs))
        e_orc = rms(got[0] - ref_orc)
        record("forward_pipeline_" + ("batched_gru" if batched else "default"), worst_rel_rms_vs_plain_forward=worst,
               rms_err_vs_oracle=e_orc)
        # default: the very same kernels -> identical;  batched GRU: two fp32-class recurrences differ in rounding
        assert worst <= (2e-5 if batched else 0.0), (batched, worst)
        assert e_orc <= 1e-4
    with pytest.raises(ValueError):
        nws_amd.ForwardPipeline(fast, depth=1)
    # default RNG path: draws come from the device generator in submit order
    torch.manual_seed(123)
    a1 = pipe.submit(jobs[0][0], jobs[0][1])
    pipe.synchronize()
    torch.manual_seed(123)
    a2 = pipe.submit(jobs[0][0], jobs[0][1])
    pipe.synchronize()
    assert torch.equal(a1, a2)


def test_forward_pipeline_soak_bench_shape_with_skews(models):
    """The bench's own shape (64 x 500), 3 x 16 distinct batches through the default pipeline, with extra work injected into
    the streams at random to shift their relative timing: every output must be bit-identical to model()'s.  (This is the
    check that exposed the two-audio-stream problem described in pipeline.py.)"""
    import ctypes as C
    import random
    import nws_amd
    _lib = nws_amd._lib
    _, fast = models
    eng = fast._engine
    w, _, _ = eng.weights()
    B, T = 64, 500
    g = torch.Generator(device="cuda").manual_seed(0)
    jobs = []
    for _ in range(16):
        f0 = (100 + 900 * torch.rand(B, 1, 1, device="cuda", generator=g)) * (1 + 0.01 * torch.randn(B, 1, T, device="cuda", generator=g))
        jobs.append((f0, torch.randn(B, 2, T, device="cuda", generator=g), torch.rand(101, device="cuda", generator=g),
                     torch.rand(128 * T - 1, device="cuda", generator=g)))
    refs = [fast(f0, c, phase_u=pu, noise=nz) for f0, c, pu, nz in jobs]
    torch.cuda.synchronize()
    scratch = torch.empty(B, T, 128, device="cuda")

    def skew(stream, n):
        with torch.cuda.stream(stream):
            for _ in range(n):
                _lib.check(_lib.lib().nws_control_gru(C.byref(w), _lib.ptr(jobs[0][1]), B, 2, T, _lib.ptr(scratch),
                                                      _lib.stream_ptr()), "nws_control_gru")
    rng = random.Random(1)
    bad = 0
    for kw in (dict(), dict(depth=4, control_streams=2)):
        pipe = nws_amd.ForwardPipeline(fast, **kw)
        for rep in range(3):
            outs = []
            for f0, c, pu, nz in jobs:
                if rep and rng.random() < 0.5:
                    skew(rng.choice(pipe.control + pipe.audio), rng.choice([1, 1, 2]))
                outs.append(pipe.submit(f0, c, phase_u=pu, noise=nz))
            pipe.synchronize()
            bad += sum(0 if torch.equal(o, r) else 1 for o, r in zip(outs, refs))
    record("forward_pipeline_soak", batches=2 * 3 * len(jobs), mismatching=bad)
    assert bad == 0


@pytest.mark.parametrize("B,T", [(40, 300), (16, 7), (64, 500)])
def test_gru_batched_mfma_path(models, weights, B, T):
    """nws_control_gru_batched runs the recurrence as a GEMM per step on the matrix cores (16 utterances per workgroup, fp16
    two-term split).  Held to the same yard-stick as the per-utterance kernel: a float64 GRU, error class of torch's own fp32
    GRU; plus the carried-state form (two halves == one pass)."""
    import ctypes as C
    import nws_amd
    _lib = nws_amd._lib
    check, ptr, stream_ptr = _lib.check, _lib.ptr, _lib.stream_ptr
    m, _ = models
    eng = m._engine
    g = torch.Generator().manual_seed(B * 1000 + T)
    control = torch.randn(B, 3, T, generator=g) * torch.linspace(0.2, 3.0, B).view(B, 1, 1)
    got = eng.control_gru(control.cuda(), batched=True).cpu().numpy()
    small = eng.control_gru(control.cuda()).cpu().numpy()
    g64 = _gru_float64(weights, control.numpy())
    gru_t = torch.nn.GRU(2, 128, batch_first=True)
    with torch.no_grad():
        for name in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            getattr(gru_t, name).copy_(torch.from_numpy(weights["embedding.gru." + name]))
        ref32 = gru_t(control[:, :2].transpose(1, 2))[0].numpy()
    e64, e_small64, e_ref64 = maxabs(got, g64), maxabs(small, g64), maxabs(ref32, g64)
    record(f"gru_mfma_B{B}_T{T}", vs_f64=e64, per_utterance_kernel_vs_f64=e_small64, torch_vs_f64=e_ref64,
           vs_per_utterance_kernel=maxabs(got, small))
    assert e64 <= max(5e-6, 3.0 * e_ref64), (e64, e_ref64)
    # carried state: two halves through nws_control_gru_state == one pass
    w, _, dev_ = eng.weights()
    Ta = T // 2 + 1
    h = torch.zeros(B, 128, device="cuda")
    outs = []
    for a, b_ in ((0, Ta), (Ta, T)):
        if b_ <= a:
            continue
        c = control[:, :, a:b_].contiguous().cuda()
        o = torch.empty(B, b_ - a, 128, device="cuda")
        hn = torch.empty_like(h)
        check(_lib.lib().nws_control_gru_batched(C.byref(w), ptr(c), B, 3, b_ - a, ptr(h), ptr(o), ptr(hn), stream_ptr()),
              "nws_control_gru_batched")
        outs.append(o)
        h = hn
    two = torch.cat(outs, dim=1).cpu().numpy()
    assert maxabs(two, got) <= 1e-6
    assert maxabs(h.cpu().numpy(), got[:, -1]) <= 1e-6


def test_frame_mlps_fp32_fallback_kernel(models, oracle):
    """The exact-fp32 MFMA kernel (used when weight norms could overflow the fp16 two-term split) stays correct."""
    m = build_model(False)
    m._engine.fp16_mlp_safe = lambda *a, **k: False
    m.invalidate_cache()
    assert not m._engine.weights()[0].mlp_frags
    g = load_npz("g1_realistic.npz")
    st = {}
    oracle[0](g["f0"], g["control"], g["phase_u"], g["noise"], stages=st)
    emb, film, H, fir = m._engine.frame_mlps(st["gru_out"].contiguous().cuda(), want_emb=True, want_H=True)
    assert maxabs(emb.cpu().numpy(), st["embedding"].numpy()) <= 1e-5
    assert maxabs(film.cpu().numpy(), st["film"].numpy().transpose(0, 2, 1)) <= 5e-5
    assert maxabs(H.cpu().numpy(), st["H"].numpy().transpose(0, 2, 1)) <= 5e-5
    y = m(dev(g["f0"]), dev(g["control"]), phase_u=dev(g["phase_u"]), noise=dev(g["noise"])).cpu().numpy()
    assert rms(y - g["y_newt"]) <= 1e-4


def test_fp16_mlp_guard_trips_on_real_weights(weights):
    """Engine.fp16_mlp_safe with weights that really break the fp16 two-term split: the h_generator's last LayerNorm gain is
    scaled so that its outputs (|gamma| sqrt(C-1) ~ 2e5) leave fp16 range.  The guard must pick the exact-fp32 kernel by
    itself, and the result must match the oracle run on the SAME modified weights."""
    from oracle.newt_oracle import OracleNEWT
    w2 = {k: np.array(v, copy=True) for k, v in weights.items()}
    w2["h_generator.net.7.layer_norm.weight"] = w2["h_generator.net.7.layer_norm.weight"] * 2.0e4
    w2["h_generator.net.9.weight"] = w2["h_generator.net.9.weight"] / 2.0e4          # keep H (hence the audio) in a sane range
    m = build_model(False)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in w2.items()})
    m = m.cuda()
    assert not m._engine.fp16_mlp_safe()                      # the guard itself, no monkey-patching
    assert not m._engine.weights()[0].mlp_frags               # -> exact-fp32 MFMA kernel
    assert build_model(False)._engine.weights()[0].mlp_frags  # the shipped weights take the fp16 path
    g = load_npz("g1_realistic.npz")
    ref = OracleNEWT(w2, fast=False)(g["f0"], g["control"], g["phase_u"], g["noise"]).numpy()
    y = m(dev(g["f0"]), dev(g["control"]), phase_u=dev(g["phase_u"]), noise=dev(g["noise"])).cpu().numpy()
    e = rms(y - ref)
    record("fp16_mlp_guard_tripped", rms_err=e, out_rms=rms(ref))
    assert e <= 1e-4, e


def test_fir_noise_stage(models, oracle):
    m, _ = models
    g1 = load_npz("g1_realistic.npz")
    for name in ("g3_stages.npz", "g1_realistic.npz"):
        g = load_npz(name)
        d = g if "phase_u" in g else g1
        st = {}
        oracle[0](g["f0"], g["control"], d["phase_u"], d["noise"], stages=st)
        Ht = st["H"].transpose(1, 2)
        h = (torch.fft.irfft(torch.complex(Ht, torch.zeros_like(Ht))).roll(128, -1) * torch.hann_window(256).view(1, 1, -1)).contiguous()
        h = h[..., 128:].contiguous()                       # upper half-taps, the layout nws_fir_noise takes
        out = m._engine.fir_noise(h.cuda(), dev(d["noise"]))
        err = maxabs(out.cpu().numpy(), st["noise_out"].numpy())
        scale = float(np.abs(st["noise_out"].numpy()).max())
        record("fir_noise_" + name, max_abs_err=err, signal_max=scale)
        assert err <= 2e-6 * max(scale, 1e-3) + 1e-7, (name, err, scale)
        add = torch.randn_like(out)
        out2 = m._engine.fir_noise(h.cuda(), dev(d["noise"]), add_in=add)
        assert maxabs(out2.cpu().numpy(), (add + out).cpu().numpy()) <= 1e-6


@pytest.mark.parametrize("B,T", [(40, 9), (17, 3), (64, 33)])
def test_fir_noise_batched_mfma_path(models, oracle, B, T):
    """B >= 16 runs the shared-noise circulant GEMM on the matrix cores (fp16 two-term split).  Checked against the oracle's
    STFT/iSTFT formulation and against the packed-fp32 kernel that serves small batches (run here in slices of 8)."""
    m, _ = models
    eng = m._engine
    g = torch.Generator().manual_seed(100 * B + T)
    H = (0.02 * torch.rand(B, 129, T, generator=g) ** 3 + 1e-4)
    H[0] *= 50.0                                    # one loud utterance, one (nearly) silent: the tap scale must not matter
    H[1] *= 1e-4
    noise = torch.rand(128 * T - 1, generator=g)
    ref = oracle[0].fir_noise(H, noise)[:, 0].numpy()
    Ht = H.transpose(1, 2)
    h = (torch.fft.irfft(torch.complex(Ht, torch.zeros_like(Ht))).roll(128, -1) * torch.hann_window(256).view(1, 1, -1))[..., 128:].contiguous().cuda()
    add = torch.randn(B, 128 * T, generator=g).cuda()
    out = eng.fir_noise(h, noise.cuda()).cpu().numpy()
    small = torch.cat([eng.fir_noise(h[i:i + 8].contiguous(), noise.cuda()) for i in range(0, B, 8)]).cpu().numpy()
    rowmax = np.abs(ref).max(axis=1)
    err_rows = np.abs(out - ref).max(axis=1)
    record(f"fir_noise_mfma_B{B}_T{T}", max_rel_row_err=float((err_rows / np.maximum(rowmax, 1e-12)).max()),
           vs_small_kernel=maxabs(out, small), signal_max=float(rowmax.max()))
    assert np.all(err_rows <= 2e-6 * rowmax + 1e-12), (err_rows / rowmax).max()
    assert np.all(np.abs(out - small).max(axis=1) <= 2e-6 * rowmax + 1e-12)
    out2 = eng.fir_noise(h, noise.cuda(), add_in=add).cpu().numpy()
    assert maxabs(out2, add.cpu().numpy() + out) <= 1e-6


# 256, 640, 1024: the time-domain form of short buffers (csrc/reverb_fft.hip reverb_direct_kernel); 1152 and up: the FFT -
# direct plans with 125 columns (32000, 64000, 128 x 2000 = 125 x 2048) or a DFT-matrix column pass (128 x 504 = 63 x 1024,
# 128 x 512 = 32 x 2048), and overlap-save plans for every other length: 128 x 251 (two blocks of 64000 whose history wraps
# around Lc = 32128 twice), 128 x 501 (one block of 128000), 128 x 8193 = 65.5 s and 128 x 9375 = 75 s (blocks of 256000)
@pytest.mark.parametrize("N", [256, 640, 1024, 1152, 4096, 32000, 128 * 251, 64000, 128 * 501, 128 * 504, 128 * 512, 128 * 2000,
                               128 * 8193, 128 * 9375])
def test_reverb_stage(models, oracle, N):
    m, _ = models
    torch.manual_seed(N)
    x = torch.randn(3, N)
    ref = oracle[0].reverb(x).numpy()
    y = m._engine.reverb(x.cuda()).cpu().numpy()
    err = rms(y - ref)
    record(f"reverb_N{N}", rms_err=err, ref_rms=rms(ref))
    assert err <= 3e-6 * rms(ref), (err, rms(ref))   # both sides are fp32 FFTs; relative 1e-6 class
    y2 = m.reverb(x.cuda()).cpu().numpy()             # stand-alone module forward uses the same kernels
    assert np.array_equal(y, y2)


@pytest.mark.parametrize("n2", [256, 512, 1024, 2048])
def test_reverb_overlap_save_block_sizes(models, oracle, n2, monkeypatch):
    """Every block size of the overlap-save plan (125 x 256 .. 125 x 2048 points; NWS_REVERB_OLS_N2 pins what the cost model
    would choose) against torch's FFT result, on a length whose history wraps (N = 128 x 1001: Lc = N)."""
    import ctypes as C
    from nws_amd import _lib
    if n2 == 256:
        pytest.skip("125 x 256 = 32000 points cannot hold 31999 samples of history plus output")
    monkeypatch.setenv("NWS_REVERB_OLS_N2", str(n2))
    m, _ = models
    N = 128 * 1001
    plan = _lib.NwsReverbPlan()
    assert _lib.lib().nws_reverb_plan(N, 32000, C.byref(plan)) == 0
    assert (plan.N1, plan.N2, plan.Lc, plan.hist) == (125, n2, N, 31999) and plan.nblk == -(-N // (125 * n2 - 31999))
    from nws_amd import engine as nws_engine
    nws_engine._PLAN_CACHE.clear()
    try:
        torch.manual_seed(n2)
        x = torch.randn(5, N)
        ref = oracle[0].reverb(x).numpy()
        y = m._engine.reverb(x.cuda()).cpu().numpy()
    finally:
        nws_engine._PLAN_CACHE.clear()
    err = rms(y - ref)
    record(f"reverb_ols_n2_{n2}", rms_err=err, ref_rms=rms(ref))
    assert err <= 3e-6 * rms(ref), (err, rms(ref))


@pytest.mark.parametrize("T", [8193, 9375, 10001, 37500])
def test_long_clips_render_and_match_the_oracle(models, oracle, T):
    """One-shot forwards of lengths the round-3 reverb plan refused (65.5 s, 75 s, 80 s, 5 min - Colab cell 18
    renders whole files): B = 2, FastNEWT, row 1 against the oracle's forward (row 0 differs: its F0 glides), <= 1e-4 RMS."""
    m, o = models[1], oracle[1]
    g = torch.Generator().manual_seed(T)
    t = torch.arange(T, dtype=torch.float32) / 125.0
    f0 = torch.stack([220.0 * 2.0 ** (t / t[-1]), 330.0 + 6.0 * torch.sin(2 * np.pi * 5.5 * t)])[:, None, :]
    control = torch.randn(2, 2, T, generator=g).cumsum(-1) * 0.05
    control = (control - control.mean(-1, keepdim=True)) / control.std(-1, keepdim=True)
    pu, nz = torch.rand(101, generator=g), torch.rand(128 * T - 1, generator=g)
    with torch.no_grad():
        y = m(f0.cuda(), control.cuda(), phase_u=pu.cuda(), noise=nz.cuda()).cpu().numpy()
    assert y.shape == (2, 128 * T) and np.isfinite(y).all()
    ref = o(f0[1:2], control[1:2], pu, nz).numpy()
    err = rms(y[1:2] - ref)
    record(f"long_clip_T{T}", rms_err=err, out_rms=rms(ref), seconds=T * 128 / 16000.0)
    assert err <= 1e-4 and err <= 2e-4 * rms(ref), (T, err, rms(ref))
    assert rms(y[0]) > 1e-3 and rms(y[0] - y[1]) > 1e-3


def test_lut_table_and_lookup(models, oracle):
    exact, fast = models
    g = load_npz("g5_lut.npz")
    table = fast.newt.lookup_table.detach().cpu().numpy()
    ref_table = oracle[1].lookup_table().numpy()
    err = maxabs(table, ref_table)
    record("lut_table", max_abs_err=err)
    # 25 sins + 144 MACs per entry; sine arguments reach ~1e2 rad where one fp32 ulp of the argument is ~4e-6,
    # so FMA-vs-separate rounding inside the 8-term sums moves an entry by a few 1e-6
    assert err <= 1e-5
    assert maxabs(table[g["rows"]], g["table_rows"]) <= 1e-5
    # lookup arithmetic is bit-exact given the same table: load the reference's own rows
    import nws_amd as nws
    probe_model = build_model(True)
    with torch.no_grad():
        probe_model.newt.lookup_table.copy_(torch.from_numpy(ref_table).cuda())
    probe_model.invalidate_cache()
    xp = dev(np.broadcast_to(g["probes"], (1, 64, g["probes"].size)).copy())
    out = probe_model.newt.shaping_fn(xp).cpu().numpy()[0]
    assert np.array_equal(out, g["probe_out"])
    x = torch.randn(2, 64, 1000) * 1.5
    assert np.array_equal(probe_model.newt.shaping_fn(x.cuda()).cpu().numpy(), oracle[1].lut_shaper(x).numpy())
    # exact shapers
    ye = exact.newt.shaping_fn(x.cuda()).cpu().numpy()
    e2 = maxabs(ye, oracle[0].exact_shaper(x).numpy())
    record("exact_shaper", max_abs_err=e2)
    assert e2 <= 2e-5


def _e2e(models, oracle, g, d, tag, tol=1e-4):
    exact, fast = models
    f0, control = dev(g["f0"]), dev(g["control"])
    pu, nz = dev(d["phase_u"]), dev(d["noise"])
    y = exact(f0, control, phase_u=pu, noise=nz).cpu().numpy()
    yf = fast(f0, control, phase_u=pu, noise=nz).cpu().numpy()
    e, ef = rms(y - g["y_newt"]), rms(yf - g["y_fast"])
    record("e2e_" + tag, rms_err_exact=e, rms_err_fast=ef, out_rms=rms(g["y_newt"]))
    assert y.shape == g["y_newt"].shape
    assert e <= tol, (tag, "exact", e)
    assert ef <= tol, (tag, "fast", ef)


def test_e2e_g1_realistic(models, oracle):
    g = load_npz("g1_realistic.npz")
    _e2e(models, oracle, g, g, "g1_realistic")


def test_e2e_g2_timing_script_inputs(models, oracle):
    _e2e(models, oracle, load_npz("g2_rand.npz"), load_npz("g1_realistic.npz"), "g2_rand")


def test_e2e_g6_high_f0(models, oracle):
    _e2e(models, oracle, load_npz("g6_highf0.npz"), load_npz("g1_realistic.npz"), "g6_highf0")


@pytest.mark.parametrize("inst", ["fl", "tpt"])
def test_e2e_g7_other_instruments(inst):
    """flute / trumpet checkpoints (reference checkpoints/nws/{fl,tpt}/last.ckpt) against the reference's own outputs,
    exact shapers and FastNEWT, default kernel options and the one-term-sine option"""
    import nws_amd as nws
    from conftest import ROOT
    import os

    nws.ensure_default_config()
    path = os.path.join(ROOT, "tests", "golden", f"weights_{inst}.npz")
    exact = nws.NeuralWaveshaping.load_from_checkpoint(path).cuda().eval()
    fast = nws.NeuralWaveshaping.load_from_checkpoint(path).cuda().eval()
    fast.newt = nws.FastNEWT(fast.newt)
    g = load_npz(f"g7_{inst}.npz")
    _e2e((exact, fast), None, g, g, f"g7_{inst}")
    assert fast._engine.exciter_opts() == 0          # shipped checkpoints: the worst-case bound keeps every product two-term
    for opts, tag, tol in ((0, "two_term", 2e-6), (4, "hybrid", 1e-5), (8, "hybrid_w", 1e-5), (2, "one_term", 1e-4), (1, "valu_film", 2e-6)):
        fast.exciter_opts = opts
        fast.invalidate_cache()
        y = fast(dev(g["f0"]), dev(g["control"]), phase_u=dev(g["phase_u"]), noise=dev(g["noise"])).cpu().numpy()
        e = rms(y - g["y_fast"])
        record(f"e2e_g7_{inst}_{tag}", rms_err_fast=e, out_rms=rms(g["y_fast"]))
        assert e <= tol, (inst, tag, e)


def test_exciter_options_on_golden_vectors(models):
    """NwsWeights.exciter_opts: two-term sines (FiLM on the matrix pipe / round-1 VALU form), the hybrid the engine selects
    for the shipped checkpoints (held to 1e-5, a tenth of the bar) and pure one-term sines, against the reference's outputs
    (G1 realistic, G2 timing-script inputs, G6 high F0)."""
    _, fast = models
    d = load_npz("g1_realistic.npz")
    try:
        for name in ("g1_realistic", "g2_rand", "g6_highf0"):
            g = load_npz(name + ".npz")
            for opts, tag, tol in ((0, "two_term", 2e-6), (1, "valu_film", 2e-6), (4, "hybrid", 1e-5), (8, "hybrid_w", 1e-5),
                                   (2, "one_term", 1e-4)):
                fast.exciter_opts = opts
                fast.invalidate_cache()
                y = fast(dev(g["f0"]), dev(g["control"]), phase_u=dev(d["phase_u"]), noise=dev(d["noise"])).cpu().numpy()
                e = rms(y - g["y_fast"])
                record(f"exciter_opts_{name}_{tag}", rms_err_fast=e)
                assert e <= tol, (name, tag, e)
    finally:
        fast.exciter_opts = None
        fast.invalidate_cache()


def test_lightning_style_ckpt_loads_on_the_gpu_box(weights, tmp_path):
    """The reference's users hand `load_from_checkpoint` a pytorch-lightning 1.2.8 `.ckpt` (scripts/resynthesise_dataset.py:47,
    colab cell 6): a torch zip-pickle with `state_dict`, `hyper_parameters` and a `callbacks` dict keyed by the
    ModelCheckpoint CLASS - the one global of the pickle that needs Lightning.  The shipped .ckpt files cannot travel to
    the GPU box, so an equivalent file is written here from the weights fixture (with a throw-away stand-in class at
    pickling time, removed again before loading) and read through checkpoint.read_checkpoint's zip-pickle branch."""
    import sys
    import types

    import nws_amd as nws

    names = ["pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.callbacks.model_checkpoint"]
    assert not any(n in sys.modules for n in names)                      # no Lightning on the box
    mods = {n: types.ModuleType(n) for n in names}
    ModelCheckpoint = type("ModelCheckpoint", (), {"__module__": names[2]})
    mods[names[2]].ModelCheckpoint = ModelCheckpoint
    path = str(tmp_path / "last.ckpt")
    sys.modules.update(mods)
    try:
        torch.save({"epoch": 3, "global_step": 120000, "pytorch-lightning_version": "1.2.8",
                    "callbacks": {ModelCheckpoint: {"best_model_score": torch.tensor(1.0), "best_model_path": "x.ckpt"}},
                    "optimizer_states": [], "lr_schedulers": [],
                    "state_dict": {k: torch.as_tensor(v) for k, v in weights.items()},
                    "hyper_parameters": {"n_waveshapers": 64, "control_hop": 128, "sample_rate": 16000, "learning_rate": 1e-3,
                                         "lr_decay": 0.9, "lr_decay_interval": 10000, "log_audio": False}}, path)
    finally:
        for n in names:
            sys.modules.pop(n, None)
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=False)      # the pickle really needs the missing class ...
    nws.ensure_default_config()
    m = nws.NeuralWaveshaping.load_from_checkpoint(path).cuda().eval()     # ... and the front end supplies it
    assert not any(n in sys.modules for n in names)
    assert m.hparams["n_waveshapers"] == 64 and m.hparams["control_hop"] == 128
    for k, v in weights.items():
        assert np.array_equal(m.state_dict()[k].cpu().numpy(), v), k
    g = load_npz("g1_realistic.npz")
    y = m(dev(g["f0"]), dev(g["control"]), phase_u=dev(g["phase_u"]), noise=dev(g["noise"])).cpu().numpy()
    assert rms(y - g["y_newt"]) <= 1e-4


def _fast_model_from(w2):
    import nws_amd as nws

    m = build_model(False)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in w2.items()})
    m = m.cuda().eval()
    m.newt = nws.FastNEWT(m.newt)
    return m


def test_hybrid_w_guard_refuses_checkpoint_the_energy_rule_admitted(weights):
    """The automatic precision choice (Engine.exciter_opts) is a worst-case bound from the weights
    (precision.hybrid_w_error_bound), not round 2's share of the mixer's weight ENERGY in harmonics 1..15.  Crafted
    checkpoint: the vn weights with reverb.ir x 30.  The mixer is untouched, so the energy rule would still have selected the
    fp16 x fp16 products for harmonics 16..101 (share 61 % >= 55 %) - but every error before the reverb now reaches the
    output 30 x larger, and on the timing-script inputs (all 101 harmonics live) the hybrid-W arithmetic misses the 1e-4 bar
    against the oracle run on the SAME weights.  The guard must refuse it by itself; the two-term result stays inside."""
    from oracle.newt_oracle import OracleNEWT

    w2 = {k: np.array(v, copy=True) for k, v in weights.items()}
    w2["reverb.ir"] = w2["reverb.ir"] * 30.0
    e_mix = (w2["harmonic_mixer.weight"].astype(np.float64) ** 2).sum(axis=(0, 2))
    assert e_mix[:15].sum() / e_mix.sum() >= 0.55                       # what round 2's rule looked at: it admitted this one
    m = _fast_model_from(w2)
    b = m._engine.hybrid_w_bound()
    assert b["bound"] > 1e-5 and m._engine.exciter_opts() == 0          # refused, no monkey-patching
    assert m._engine.weights()[0].exciter_opts == 0
    g, d = load_npz("g2_rand.npz"), load_npz("g1_realistic.npz")
    ref = OracleNEWT(w2, fast=True, lut_python_loop=False)(g["f0"], g["control"], d["phase_u"], d["noise"]).numpy()
    args = (dev(g["f0"]), dev(g["control"]))
    kw = dict(phase_u=dev(d["phase_u"]), noise=dev(d["noise"]))
    y_auto = m(*args, **kw).cpu().numpy()
    m.exciter_opts = 8
    m.invalidate_cache()
    y_forced = m(*args, **kw).cpu().numpy()
    e_auto, e_forced = rms(y_auto - ref), rms(y_forced - ref)
    record("hybrid_w_guard_refused", rms_err_auto_two_term=e_auto, rms_err_forced_hybrid_w=e_forced, out_rms=rms(ref),
           bound=b["bound"], max_abs_forced_vs_auto=maxabs(y_forced, y_auto))
    assert e_auto <= 2e-5, e_auto                                       # 30 x the two-term error of the plain checkpoint
    assert e_forced > 3.0 * e_auto, (e_forced, e_auto)                  # what the refusal avoided
    assert maxabs(y_forced, y_auto) <= b["bound"]                       # the bound is a bound


def test_hybrid_w_guard_admits_when_the_bound_allows(weights):
    """...and the same rule admits EXCITER_HYBRID_W where the bound proves it harmless: (a) harmonics 16..101 with zero mixer
    weights (bound 0: the dropped terms are exact zeros, output bit-identical to the two-term kernel); (b) small gains all
    along the chain (bound <= 1e-5), checked against the oracle on the same weights."""
    from oracle.newt_oracle import OracleNEWT

    g, d = load_npz("g2_rand.npz"), load_npz("g1_realistic.npz")
    args = (dev(g["f0"]), dev(g["control"]))
    kw = dict(phase_u=dev(d["phase_u"]), noise=dev(d["noise"]))
    # (a)
    w2 = {k: np.array(v, copy=True) for k, v in weights.items()}
    w2["harmonic_mixer.weight"][:, 15:] = 0.0
    m = _fast_model_from(w2)
    assert m._engine.hybrid_w_bound()["bound"] == 0.0 and m._engine.exciter_opts() == 8
    y8 = m(*args, **kw)
    m.exciter_opts = 0
    m.invalidate_cache()
    assert torch.equal(y8, m(*args, **kw))
    # (b)
    w3 = {k: np.array(v, copy=True) for k, v in weights.items()}
    w3["harmonic_mixer.weight"][:, 15:] *= 1e-4
    w3["reverb.ir"] = w3["reverb.ir"] * 1e-3
    for k in ("newt.mlp.net.9.weight", "newt.mlp.net.9.bias"):
        w3[k] = w3[k] * 0.005
    m = _fast_model_from(w3)
    b = m._engine.hybrid_w_bound()
    assert b["bound"] <= 1e-5 and m._engine.exciter_opts() == 8, b
    ref = OracleNEWT(w3, fast=True, lut_python_loop=False)(g["f0"], g["control"], d["phase_u"], d["noise"]).numpy()
    y8 = m(*args, **kw).cpu().numpy()
    m.exciter_opts = 0
    m.invalidate_cache()
    y0 = m(*args, **kw).cpu().numpy()
    record("hybrid_w_guard_admitted", rms_err_hybrid_w=rms(y8 - ref), rms_err_two_term=rms(y0 - ref), out_rms=rms(ref),
           bound=b["bound"], max_abs_hybrid_vs_two_term=maxabs(y8, y0))
    assert maxabs(y8, y0) <= b["bound"] + 1e-7        # + the two kernels' own fp32 summation-order noise
    assert rms(y8 - ref) <= 1e-5


def test_e2e_g3_batch2_extra_control_channels(models, oracle):
    g = load_npz("g3_stages.npz")
    gg = dict(g, y_newt=g["y_exact"], y_fast=g["y_lut"])
    _e2e(models, oracle, gg, g, "g3_stages")


@pytest.mark.parametrize("T", [2, 32])
def test_e2e_g4_streaming_buffers(models, oracle, T):
    g = load_npz("g4_stream.npz")
    gg = dict(f0=g[f"f0_T{T}"], control=g[f"control_T{T}"], y_newt=g[f"y_newt_T{T}"], y_fast=g[f"y_fast_T{T}"])
    dd = dict(phase_u=g[f"phase_u_T{T}"], noise=g[f"noise_T{T}"])
    _e2e(models, oracle, gg, dd, f"g4_T{T}")


def test_batch64_rows_match_single_and_oracle(models, oracle):
    """Full-size batch (config 3 shape): every row equals its own B=1 render; spot-check rows vs the oracle."""
    _, fast = models
    torch.manual_seed(3)
    B, T = 64, 500
    f0 = (100 + 900 * torch.rand(B, 1, 1) * (1 + 0.01 * torch.sin(torch.linspace(0, 40, T)).view(1, 1, T))).contiguous()
    control = torch.randn(B, 2, T)
    pu, nz = torch.rand(101), torch.rand(128 * T - 1)
    y = fast(f0.cuda(), control.cuda(), phase_u=pu.cuda(), noise=nz.cuda())
    for r in (0, 17, 63):
        y1 = fast(f0[r:r + 1].cuda(), control[r:r + 1].cuda(), phase_u=pu.cuda(), noise=nz.cuda())
        assert torch.equal(y1[0], y[r]) or rms((y1[0] - y[r]).cpu().numpy()) <= 1e-6
        ref = oracle[1](f0[r:r + 1], control[r:r + 1], pu, nz).numpy()
        e = rms(y[r:r + 1].cpu().numpy() - ref)
        record(f"b64_row{r}", rms_err=e, out_rms=rms(ref))
        assert e <= 1e-4


@pytest.mark.parametrize("B,T", [(1, 2), (3, 31), (15, 33), (16, 64), (17, 65), (33, 129), (130, 70), (70, 257)])
def test_shape_sweep_matches_the_oracle(models, oracle, B, T):
    """Batch / frame counts on both sides of every kernel-selection threshold of the fused path: fewer than 16 utterances (the
    per-utterance noise kernel) and more (the shared-noise MFMA form, with partial 32-utterance tiles: 17, 33, 130), fewer than
    256 frame tiles (32-frame MLP kernel) and more (64-frame tiles with a partial last tile: T = 70, 257), odd batch sizes (the
    reverb packs two utterances per transform), circular lengths 32 000 (row pass of 256 points) and 32 896 = 128 x 257 (the MFMA
    column pass), the time-domain reverb of short buffers (T = 2).  Three rows of each against the oracle, exact and FastNEWT;
    rows with F0 = 0 and F0 beyond Nyquist ride along."""
    g = torch.Generator().manual_seed(1000 * B + T)
    f0 = 80.0 + 1500.0 * torch.rand(B, 1, 1, generator=g) * (0.5 + torch.rand(B, 1, T, generator=g))
    f0[0, 0, : T // 2] = 0.0
    f0[-1, 0, T // 2:] = 9000.0
    control = torch.randn(B, 2, T, generator=g)
    pu, nz = torch.rand(101, generator=g), torch.rand(128 * T - 1, generator=g)
    rows = sorted({0, B // 2, B - 1})
    for which, (m, o) in enumerate(zip(models, oracle)):
        with torch.no_grad():
            y = m(f0.cuda(), control.cuda(), phase_u=pu.cuda(), noise=nz.cuda()).cpu().numpy()
        assert y.shape == (B, 128 * T)
        ref = o(f0[rows], control[rows], pu, nz).numpy()
        errs = [rms(y[r] - ref[i]) for i, r in enumerate(rows)]
        record(f"shape_sweep_B{B}_T{T}_{'fast' if which else 'exact'}", rms_err_max=max(errs), out_rms=rms(ref))
        assert max(errs) <= 1e-4 and max(errs) <= 2e-5 * max(rms(ref), 1e-3), (B, T, which, errs, rms(ref))


@pytest.mark.parametrize("size,lo,hi", [(1000, -2.5, 3.5), (4096, -4.0, 4.0), (512, -3.0, 3.0), (4096, -3.0, 3.0)])
def test_fastnewt_table_parameters(weights, size, lo, hi):
    """FastNEWT's gin-configurable table (shaping.py:82-105: table_size, table_min, table_max) on the FUSED path: only a
    power-of-two size over a range of exactly 6 takes the folded-index kernel, everything else the general pair-table kernel;
    both against the oracle built with the same three numbers (g1's inputs reach outside [-2.5, 3.5]: the below-range
    extrapolation and the flat top of the reference's lookup are part of the comparison)."""
    import nws_amd as nws
    from oracle.newt_oracle import OracleNEWT

    g = load_npz("g1_realistic.npz")
    m = build_model(False)
    m.newt = nws.FastNEWT(m.newt, table_size=size, table_min=lo, table_max=hi)
    assert m._engine.specialised()
    ref = OracleNEWT(weights, fast=True, lut_python_loop=False, table_size=size, table_min=lo, table_max=hi)(
        g["f0"], g["control"], g["phase_u"], g["noise"]).numpy()
    with torch.no_grad():
        y = m(dev(g["f0"]), dev(g["control"]), phase_u=dev(g["phase_u"]), noise=dev(g["noise"])).cpu().numpy()
    e = rms(y - ref)
    record(f"fastnewt_table_{size}_{lo}_{hi}", rms_err=e, out_rms=rms(ref))
    assert e <= 1e-4 and e <= 2e-4 * rms(ref), (size, lo, hi, e, rms(ref))


@pytest.mark.parametrize("sr", [22050, 8000])
def test_fused_path_at_other_sample_rates(weights, sr):
    """sample_rate is a gin parameter of NeuralWaveshaping and HarmonicOscillator (neural_waveshaping.py:35, generators.py:40-44):
    it sets the phase increment and the anti-aliasing mask, not a size, so the fused kernels serve it; against the oracle."""
    import nws_amd as nws
    from conftest import GOLDEN
    from oracle.newt_oracle import OracleNEWT
    import os

    g = load_npz("g1_realistic.npz")
    nws.ensure_default_config()
    try:
        nws.gin.parse_config(f"NeuralWaveshaping.sample_rate = {sr}\nHarmonicOscillator.sample_rate = {sr}\n")
        for fast in (False, True):
            # hyper-parameters stored with a checkpoint win over gin, as in Lightning's load_from_checkpoint: pass the override
            m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(GOLDEN, "weights_vn.npz"), sample_rate=sr).cuda().eval()
            if fast:
                m.newt = nws.FastNEWT(m.newt)
            assert float(m.sample_rate) == sr and float(m.osc.sample_rate) == sr and m._engine.specialised()
            ref = OracleNEWT(weights, sample_rate=sr, fast=fast, lut_python_loop=False)(g["f0"], g["control"], g["phase_u"], g["noise"]).numpy()
            with torch.no_grad():
                y = m(dev(g["f0"]), dev(g["control"]), phase_u=dev(g["phase_u"]), noise=dev(g["noise"])).cpu().numpy()
            e = rms(y - ref)
            record(f"sample_rate_{sr}_{'fast' if fast else 'exact'}", rms_err=e, out_rms=rms(ref))
            assert e <= 1e-4 and e <= 2e-4 * rms(ref), (sr, fast, e, rms(ref))
    finally:
        nws.gin.clear_config()
        nws.gin.parse_config_file(nws.DEFAULT_GIN)


def test_default_rng_path_and_determinism(models):
    _, fast = models
    f0 = 220 + 50 * torch.rand(2, 1, 16, device="cuda")
    c = torch.randn(2, 2, 16, device="cuda")
    torch.manual_seed(5)
    a = fast(f0, c)
    torch.manual_seed(5)
    b = fast(f0, c)
    assert a.shape == (2, 2048) and torch.isfinite(a).all()
    assert torch.equal(a, b)
    # the draws are consumed in the reference's order and sizes from the device generator
    torch.manual_seed(5)
    pu = torch.rand_like(fast.osc.rand_phase)
    nz = torch.rand(128 * 16 - 1, device="cuda")
    assert torch.equal(fast(f0, c, phase_u=pu, noise=nz), a)
    assert not torch.equal(fast(f0, c), a)


def test_render_exciter_public_api(models, oracle):
    m, _ = models
    g = load_npz("g3_stages.npz")
    f0_up = dev(g["f0_up"]).unsqueeze(1)
    torch.manual_seed(9)
    exc = m.render_exciter(f0_up)
    torch.manual_seed(9)
    pu = torch.rand_like(m.osc.rand_phase).cpu().reshape(-1)
    ref = oracle[0].exciter(torch.from_numpy(g["f0_up"]).unsqueeze(1), pu).numpy()
    assert exc.shape == (2, 64, 384)
    assert maxabs(exc.cpu().numpy(), ref) <= 2e-5


def test_errors_are_loud(models):
    m, _ = models
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 8, device="cuda"), torch.zeros(1, 1, 8, device="cuda"))   # control needs >= 2 channels
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 8, device="cuda"), torch.zeros(2, 2, 8, device="cuda"))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 8), torch.zeros(1, 2, 8))                                   # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        m.embedding(torch.zeros(1, 3, 8, device="cuda"))                                # GRU(2 -> 128): 2 input channels
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 1, device="cuda"), torch.zeros(1, 2, 1, device="cuda"))     # one frame: reflect padding undefined


def test_hipgraph_capture_replay(models):
    """Streaming mode (config 5): the whole forward is capturable; replay == eager."""
    _, fast = models
    T = 2
    f0 = 200 + 100 * torch.rand(1, 1, T, device="cuda")
    c = torch.randn(1, 2, T, device="cuda")
    pu = torch.rand(101, device="cuda")
    nz = torch.rand(128 * T - 1, device="cuda")
    eager = fast(f0, c, phase_u=pu, noise=nz).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fast(f0, c, phase_u=pu, noise=nz)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fast(f0, c, phase_u=pu, noise=nz)
    f0.add_(5.0)
    graph.replay()
    torch.cuda.synchronize()
    ref = fast(f0, c, phase_u=pu, noise=nz)
    assert torch.equal(out, ref)
    assert not torch.equal(out, eager)


def test_exciter_wide_phase_path(models, oracle):
    """32 s at ~300 Hz: k*phase passes 6e6 rad in the last part of the clip, where the oscillator switches (per wave) from
    the packed fp32 turn reduction to the inline fp64 one.  The fp32 argument chain is reproduced exactly on both sides, so
    the mixed harmonics must still agree to sine accuracy."""
    _, fast = models
    g = torch.Generator().manual_seed(77)
    N = 128 * 4000
    f0_up = (250 + 100 * torch.rand(1, 1, N, generator=g))
    torch.manual_seed(5)
    got = fast.render_exciter(f0_up.cuda()).cpu().numpy()
    torch.manual_seed(5)
    u = torch.rand_like(fast.osc.rand_phase).cpu().reshape(-1)
    ref = oracle[0].exciter(f0_up, u).numpy()
    phase_end = 2 * np.pi * float(f0_up.double().sum()) / 16000.0
    assert phase_end * 112 > 6.0e6, phase_end          # the wide path really is exercised
    tail = slice(N - 65536, N)
    record("exciter_wide_phase", max_abs_err=maxabs(got, ref), tail_max_abs_err=maxabs(got[..., tail], ref[..., tail]),
           phase_end=phase_end)
    assert maxabs(got, ref) <= 2e-5


@pytest.mark.parametrize("B,T", [(3, 501), (1, 1000), (5, 33), (2, 250), (2, 251), (1, 2000)])
def test_e2e_odd_shapes_against_oracle(models, oracle, B, T):
    """Shapes off the beaten path: T=501 -> L=64128=501x128 (general-N1 MFMA DFT), T=1000 -> L=128000=125x1024,
    odd batch (one half-empty reverb pair), partial 32-frame MLP tiles, N<32000 (zero-padded to L=32000), T=251 -> the
    first length past the IR (L=32128=251x128, prime N1, odd hop count for the two-hops-per-workgroup oscillator),
    T=2000 -> 16 s, L=256000=250x1024."""
    exact, fast = models
    g = torch.Generator().manual_seed(1000 * B + T)
    f0 = (80 + 1500 * torch.rand(B, 1, 1, generator=g)) * (1 + 0.02 * torch.randn(B, 1, T, generator=g))
    control = torch.randn(B, 4, T, generator=g)
    pu, nz = torch.rand(101, generator=g), torch.rand(128 * T - 1, generator=g)
    for model, orc, tag in ((fast, oracle[1], "fast"), (exact, oracle[0], "exact")):
        if tag == "exact" and T > 600:
            continue  # the exact-shaper oracle is slow on 8 s clips; FastNEWT covers the shape
        y = model(f0.cuda(), control.cuda(), phase_u=pu.cuda(), noise=nz.cuda()).cpu().numpy()
        ref = orc(f0, control, pu, nz).numpy()
        e = rms(y - ref)
        record(f"odd_B{B}_T{T}_{tag}", rms_err=e, out_rms=rms(ref))
        assert y.shape == (B, 128 * T) and e <= 1e-4, (tag, e)


def test_e2e_f0_edge_cases(models, oracle):
    """F0 = 0, negative, exactly at / above Nyquist, tiny and huge: the anti-alias mask and the phase chain must agree
    with the reference semantics (fl(f0*k) < sr/2 on the upsampled F0; negative F0 keeps every harmonic)."""
    _, fast = models
    T = 24
    rows = [np.zeros(T), np.full(T, -220.0), np.full(T, 8000.0), np.full(T, 7999.5), np.full(T, 12000.0),
            np.full(T, 79.3), np.linspace(0.0, 9000.0, T), np.full(T, 3999.9999), np.full(T, 1e-3)]
    f0 = torch.tensor(np.stack(rows), dtype=torch.float32).unsqueeze(1)
    g = torch.Generator().manual_seed(4)
    control = torch.randn(len(rows), 2, T, generator=g)
    pu, nz = torch.rand(101, generator=g), torch.rand(128 * T - 1, generator=g)
    y = fast(f0.cuda(), control.cuda(), phase_u=pu.cuda(), noise=nz.cuda()).cpu().numpy()
    ref = oracle[1](f0, control, pu, nz).numpy()
    errs = [rms(y[i] - ref[i]) for i in range(len(rows))]
    record("f0_edge_cases", **{f"row{i}": e for i, e in enumerate(errs)})
    assert max(errs) <= 1e-4, errs
    exc = fast.render_exciter  # public API with a pre-upsampled F0 that is NOT a linear ramp
    f0_up = (200 + 100 * torch.rand(2, 1, 128 * 6, generator=g)).cuda()
    torch.manual_seed(11)
    e1 = exc(f0_up).cpu().numpy()
    torch.manual_seed(11)
    u = torch.rand_like(fast.osc.rand_phase).cpu().reshape(-1)
    ref_e = oracle[0].exciter(f0_up.cpu(), u).numpy()
    assert maxabs(e1, ref_e) <= 2e-5


def test_multi_stream_forwards_and_strided_inputs(models, oracle):
    """Forwards issued on different streams right after a cache rebuild must not race the derived tables; strided
    inputs (channel slices, expand) are accepted like in the reference."""
    _, fast = models
    g = torch.Generator().manual_seed(77)
    T = 40
    f0 = (150 + 300 * torch.rand(2, 1, T, generator=g)).cuda()
    big = torch.randn(2, 6, T, generator=g).cuda()
    pu, nz = torch.rand(101, generator=g).cuda(), torch.rand(128 * T - 1, generator=g).cuda()
    ref = oracle[1](f0.cpu(), big[:, 1:3].cpu(), pu.cpu(), nz.cpu()).numpy()
    fast.invalidate_cache()                          # force a rebuild of every derived table on the next call
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = []
    for i in range(6):
        with torch.cuda.stream(streams[i % 3]):
            streams[i % 3].wait_stream(torch.cuda.current_stream())
            outs.append(fast(f0, big[:, 1:3], phase_u=pu, noise=nz))   # non-contiguous channel slice
    torch.cuda.synchronize()
    for y in outs:
        assert rms(y.cpu().numpy() - ref) <= 1e-4
    y2 = fast(f0[:1, :, 0:1].expand(1, 1, T), big[:1, 1:3], phase_u=pu, noise=nz)   # expand() view of a constant F0
    ref2 = oracle[1](f0[:1, :, 0:1].expand(1, 1, T).cpu(), big[:1, 1:3].cpu(), pu.cpu(), nz.cpu()).numpy()
    assert rms(y2.cpu().numpy() - ref2) <= 1e-4


def test_offline_render_cli(tmp_path, oracle):
    """scripts/resynthesise_dataset.py (reference scripts/resynthesise_dataset.py:55-76) on a small synthetic dataset with
    ragged lengths: every wav against the oracle on the same controls and the same injected draws (--draws), target wavs
    (--write-targets), reproducibility of --seed."""
    import subprocess
    import sys
    from scipy.io import wavfile
    from conftest import ROOT
    import os

    root = tmp_path / "data"
    (root / "test" / "control").mkdir(parents=True)
    (root / "test" / "audio").mkdir(parents=True)
    w = load_npz("weights_vn.npz")
    mean, std = np.zeros((19, 1)), np.ones((19, 1))
    mean[:2, 0], std[:2, 0] = w["__data_mean__"], w["__data_std__"]
    np.save(root / "data_mean.npy", mean)
    np.save(root / "data_std.npy", std)
    rng = np.random.default_rng(1)
    lengths = [16, 16, 9, 16, 9]
    controls = {}
    for i, T in enumerate(lengths):
        c = rng.normal(size=(19, T)).astype(np.float32)
        c[0] = 0.3 * c[0]                                   # F0 around the data mean (a few hundred Hz)
        controls[f"clip{i}"] = c
        np.save(root / "test" / "control" / f"control_clip{i}.npy", c)
    target = (0.1 * rng.normal(size=16 * 128)).astype(np.float32)
    np.save(root / "test" / "audio" / "audio_clip1.npy", target)
    pu, nz = rng.random(101).astype(np.float32), rng.random(16 * 128 - 1).astype(np.float32)
    np.savez(tmp_path / "draws.npz", phase_u=pu, noise=nz)
    cli = [sys.executable, os.path.join(ROOT, "scripts", "resynthesise_dataset.py"), "--model-checkpoint",
           os.path.join(ROOT, "tests", "golden", "weights_vn.npz"), "--dataset-root", str(root), "--use-fastnewt", "--batch-size", "2"]
    out = tmp_path / "wav"
    r = subprocess.run(cli + ["--output-path", str(out), "--draws", str(tmp_path / "draws.npz"), "--write-targets"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    worst = 0.0
    for i, T in enumerate(lengths):
        sr, y = wavfile.read(out / f"clip{i}.output.wav")
        assert sr == 16000 and y.dtype == np.float32 and y.shape == (128 * T,)
        c = controls[f"clip{i}"]
        f0_hz = (c[0:1].astype(np.float64) * std[0, 0] + mean[0, 0]).astype(np.float32)        # general.py:49
        ref = oracle[1](torch.from_numpy(f0_hz)[None], torch.from_numpy(c)[None], torch.from_numpy(pu),
                        torch.from_numpy(nz[:128 * T - 1].copy())).numpy()[0]
        e = rms(y - ref)
        worst = max(worst, e)
        assert e <= 1e-4, (i, e)
    record("offline_render_cli", worst_rms_err_vs_oracle=worst, clips=len(lengths))
    sr, tgt = wavfile.read(out / "clip1.target.wav")
    assert np.array_equal(tgt, target) and not (out / "clip0.target.wav").exists()
    # --seed: reproducible, and a different seed gives different hidden draws
    runs = []
    for tag, seed in (("a", 7), ("b", 7), ("c", 8)):
        o = tmp_path / ("wav_" + tag)
        r = subprocess.run(cli + ["--output-path", str(o), "--seed", str(seed)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(np.concatenate([wavfile.read(o / f"clip{i}.output.wav")[1] for i in range(len(lengths))]))
    assert np.array_equal(runs[0], runs[1]) and not np.array_equal(runs[0], runs[2])
    assert np.isfinite(runs[2]).all() and np.abs(runs[2]).max() > 0


def test_full_size_properties(models):
    """Size-independent invariants at the bench size (B=64, T=500), where running the oracle on everything is too slow:
    linearity of the reverb and of the FIR-noise branch in its taps, shard-concatenation == un-sharded batch."""
    _, fast = models
    eng = fast._engine
    B, T = 64, 500
    N = 128 * T
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.randn(B, N, device="cuda", generator=g) * 0.01
    y = torch.randn(B, N, device="cuda", generator=g) * 0.01
    rx, ry = eng.reverb(x), eng.reverb(y)
    rc = eng.reverb(2.0 * x - 0.5 * y)
    lin = (2.0 * rx - 0.5 * ry - rc)
    scale = float(rc.pow(2).mean().sqrt())
    err = float(lin.pow(2).mean().sqrt())
    record("prop_reverb_linearity", rel=err / scale)
    assert err <= 3e-6 * scale                       # circular convolution is linear: only fp32 FFT noise remains
    fir = torch.randn(B, T, 128, device="cuda", generator=g) * 1e-3      # upper half-taps
    nz = torch.rand(N - 1, device="cuda", generator=g)
    n1 = eng.fir_noise(fir, nz)
    n2 = eng.fir_noise(-3.0 * fir, nz)
    assert float((n2 + 3.0 * n1).abs().max()) <= 1e-6 * max(1.0, float(n1.abs().max()) * 3)   # homogeneous of degree 1
    # shard-concatenation (what parallel.render_sharded does across ranks) == one un-sharded forward
    f0 = 100 + 400 * torch.rand(B, 1, T, device="cuda", generator=g)
    control = torch.randn(B, 2, T, device="cuda", generator=g)
    pu = torch.rand(101, device="cuda", generator=g)
    whole = fast(f0, control, phase_u=pu, noise=nz)
    parts = [fast(f0[i:i + 16].contiguous(), control[i:i + 16].contiguous(), phase_u=pu, noise=nz) for i in range(0, B, 16)]
    diff = float((torch.cat(parts, 0) - whole).abs().max())
    record("prop_shard_concat_maxabs", diff=diff)
    assert diff <= 2e-6   # identical arithmetic per row; only the reverb's pair packing partner changes (fp32 FFT noise)
    assert torch.isfinite(whole).all()


@pytest.mark.parametrize("seed", [0, 1])
def test_e2e_random_init_weights(seed):
    """Random-init model (what the reference's timing script runs): input_scale ~ N(0,10) drives the waveshaper
    arguments far outside the LUT range, so the below-range extrapolation / above-range clamp quirks and large
    sine arguments in the exact shapers are all exercised.  Compared against the oracle with the same weights."""
    import nws_amd as nws
    from oracle.newt_oracle import OracleNEWT

    nws.ensure_default_config()
    torch.manual_seed(seed)
    m = nws.NeuralWaveshaping()
    with torch.no_grad():
        m.reverb.ir.mul_(1e5)          # default init is 1e-6-scale: make the reverb audible
        m.newt.mlp.net[9].weight.mul_(25.0)   # large FiLM parameters: waveshaper arguments well beyond [-3, 3]
        m.newt.mlp.net[9].bias.add_(1.5)
    weights = {k: v.detach().clone().numpy() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    mf = nws.NeuralWaveshaping().cuda().eval()
    mf.load_state_dict(m.state_dict())
    mf.newt = nws.FastNEWT(mf.newt)
    g = torch.Generator().manual_seed(seed + 10)
    T = 48
    f0 = 60 + 700 * torch.rand(2, 1, T, generator=g)
    control = torch.randn(2, 2, T, generator=g)
    pu, nz = torch.rand(101, generator=g), torch.rand(128 * T - 1, generator=g)
    oe = OracleNEWT(weights, fast=False)
    of = OracleNEWT(weights, fast=True, lut_python_loop=False)
    st = {}
    ref_e = oe(f0, control, pu, nz, stages=st).numpy()
    ref_f = of(f0, control, pu, nz).numpy()
    frac_out = float((st["lut_arg"].abs() > 3).float().mean())
    ye = m(f0.cuda(), control.cuda(), phase_u=pu.cuda(), noise=nz.cuda()).cpu().numpy()
    yf = mf(f0.cuda(), control.cuda(), phase_u=pu.cuda(), noise=nz.cuda()).cpu().numpy()
    ee, ef = rms(ye - ref_e), rms(yf - ref_f)
    record(f"random_init_seed{seed}", rms_err_exact=ee, rms_err_fast=ef, out_rms_exact=rms(ref_e), out_rms_fast=rms(ref_f),
           lut_arg_fraction_outside_table=frac_out)
    assert frac_out > 0.05                                   # the case really leaves the table
    # exact shapers: sine arguments reach |x| ~ 1e2..1e3 where one fp32 ulp of the argument is ~1e-5..1e-4 rad, and the
    # oracle's conv1d rounds its 8-term sums differently from our FMA chain; bar relative to the output level
    assert ee <= 1e-4 * max(1.0, rms(ref_e)), (ee, rms(ref_e))
    # LUT: the table itself inherits that argument sensitivity; extrapolation multiplies table differences by up to
    # |fract| ~ 1e3, so compare relative to the (large) output level
    assert ef <= 2e-3 * max(1.0, rms(ref_f)), (ef, rms(ref_f))


def test_e2e_timing_script_model():
    """BASELINE config 1 LITERALLY: scripts/time_forward_pass.py:41-43 runs `NeuralWaveshaping()` with no
    checkpoint, `model.newt = FastNEWT(model.newt)`, torch.rand control / f0 at T = 500.  (a) the reference's own recorded run
    (tests/golden/g10_timing_script.npz: its random initial weights, inputs, draws, outputs for both shapers) through the product
    at the ABSOLUTE 1e-4 RMS bar of north_star; (b) the product's own unmodified random initialisation (nothing scaled, nothing
    added) against the oracle at the same absolute bar, two seeds.  The blown-up variant (test_e2e_random_init_weights) stays
    for the out-of-table quirks."""
    import nws_amd as nws
    from oracle.newt_oracle import OracleNEWT

    nws.ensure_default_config()
    z = load_npz("g10_timing_script.npz")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in z.items() if not k.startswith("__")}
    m = nws.NeuralWaveshaping()
    m.load_state_dict(sd)
    m = m.cuda().eval()
    f0, control = torch.from_numpy(z["__f0__"]).cuda(), torch.from_numpy(z["__control__"]).cuda()
    pu, nz = torch.from_numpy(z["__phase_u__"]).cuda(), torch.from_numpy(z["__noise__"]).cuda()
    ye = m(f0, control, phase_u=pu, noise=nz).cpu().numpy()
    m.newt = nws.FastNEWT(m.newt)                                      # time_forward_pass.py:43
    yf = m(f0, control, phase_u=pu, noise=nz).cpu().numpy()
    ee, ef = rms(ye - z["__y_newt__"]), rms(yf - z["__y_fast__"])
    record("timing_script_model_reference_run", rms_err_exact=ee, rms_err_fast=ef, out_rms=rms(z["__y_newt__"]))
    assert ee <= 1e-4 and ef <= 1e-4, (ee, ef)
    for seed in (0, 1):
        torch.manual_seed(seed)
        control = torch.rand(1, 2, 500)                                # :27-33
        f0 = torch.rand(1, 1, 500)                                     # :34-40
        mm = nws.NeuralWaveshaping()                                   # :41 - unmodified
        weights = {k: v.detach().clone().numpy() for k, v in mm.state_dict().items()}
        g = torch.Generator().manual_seed(100 + seed)
        pu, nz = torch.rand(101, generator=g), torch.rand(128 * 500 - 1, generator=g)
        ref_e = OracleNEWT(weights, fast=False)(f0, control, pu, nz).numpy()
        ref_f = OracleNEWT(weights, fast=True, lut_python_loop=False)(f0, control, pu, nz).numpy()
        mm = mm.cuda().eval()
        ye = mm(f0.cuda(), control.cuda(), phase_u=pu.cuda(), noise=nz.cuda()).cpu().numpy()
        mm.newt = nws.FastNEWT(mm.newt)
        yf = mm(f0.cuda(), control.cuda(), phase_u=pu.cuda(), noise=nz.cuda()).cpu().numpy()
        ee, ef = rms(ye - ref_e), rms(yf - ref_f)
        record(f"timing_script_model_own_init_seed{seed}", rms_err_exact=ee, rms_err_fast=ef, out_rms=rms(ref_e))
        assert ee <= 1e-4 and ef <= 1e-4, (seed, ee, ef)


def test_exact_shaper_bank_large_output_layer():
    """Exact mode, shaper-bank kernel: an output layer so large that its pre-activation leaves v_sin_f32's +-256-turn input
    domain (the kernel's v_fract has to bring it back).  Arguments of ~2e3 rad resolve to ~1e-4 rad in fp32 and the
    oracle's conv1d sums round differently from an FMA chain, so the bar is loose; what it must catch is zeros or garbage
    instead of sines."""
    import nws_amd as nws
    from oracle.newt_oracle import OracleNEWT

    nws.ensure_default_config()
    torch.manual_seed(3)
    m = nws.NeuralWaveshaping()
    with torch.no_grad():
        m.reverb.ir.mul_(1e5)
        m.newt.shaping_fn.net[6].weight.mul_(3000.0)      # 8 x ~0.2 x 3000 rad ~ 700 turns
    weights = {k: v.detach().clone().numpy() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    # (round 4: such a model must NOT get the kernel variant that feeds v_sin_f32 without the v_fract)
    assert not m._engine.bank_nofract_safe() and not (m._engine.weights()[0].exciter_opts & 16)
    g = torch.Generator().manual_seed(4)
    T = 24
    f0 = 100 + 500 * torch.rand(2, 1, T, generator=g)
    control = torch.randn(2, 2, T, generator=g)
    pu, nz = torch.rand(101, generator=g), torch.rand(128 * T - 1, generator=g)
    ref = OracleNEWT(weights, fast=False)(f0, control, pu, nz).numpy()
    y = m(f0.cuda(), control.cuda(), phase_u=pu.cuda(), noise=nz.cuda()).cpu().numpy()
    rel = rms(y - ref) / rms(ref)
    record("exact_bank_large_output_layer", rel_rms_err=rel, out_rms=rms(ref))
    assert np.isfinite(y).all() and rel <= 0.02, rel


def test_exact_shaper_lds_fallback_matches_bank(models, oracle):
    """The fused kernel has two exact-shaper tails: the shaper bank (weights by scalar loads from the nws_shaper_turns table,
    what the engine uses) and the fallback a C-ABI caller gets when NwsWeights.shaper_turns is NULL (weights staged in LDS
    per workgroup).  Same arithmetic in a different order: they must agree to rounding, and both with the oracle's stage."""
    import ctypes as C
    import nws_amd
    _lib = nws_amd._lib
    m, _ = models
    eng = m._engine
    w, _, dev = eng.weights()
    assert w.shaper_turns, "the engine is expected to provide the table"
    w2 = _lib.NwsWeights.from_buffer_copy(w)
    w2.shaper_turns = None
    g = torch.Generator().manual_seed(77)
    B, T = 3, 21
    f0 = (120 + 500 * torch.rand(B, 1, T, generator=g)).cuda()
    control = torch.randn(B, 2, T, generator=g).cuda()
    pu = torch.rand(101, generator=g).cuda()
    gru = eng.control_gru(control)
    _, film, _, _ = eng.frame_mlps(gru)
    carry = eng.phase_carry(f0=f0[:, 0].contiguous())
    _, bank = eng.exciter_newt(f0[:, 0].contiguous(), None, carry, pu, film)
    lds = torch.empty_like(bank)
    _lib.check(_lib.lib().nws_exciter_newt(C.byref(w2), _lib.ptr(f0[:, 0].contiguous()), None, _lib.ptr(carry), _lib.ptr(pu),
                                           _lib.ptr(eng.rand_phase()), _lib.ptr(film), B, T, float(m.sample_rate), None,
                                           _lib.ptr(lds), _lib.stream_ptr()), "nws_exciter_newt (LDS fallback)")
    torch.cuda.synchronize()
    d = float((bank - lds).abs().max())
    scale = float(bank.abs().max())
    record("exact_tail_bank_vs_lds_fallback", max_abs_diff=d, max_abs=scale)
    assert scale > 0 and d <= 2e-6 * max(1.0, scale), (d, scale)


def test_exact_shaper_bank_without_fract_matches_the_reduced_form(models):
    """Round 4: the shipped sin-MLP shapers' hidden / output pre-activations are bounded by 0.6 turns (sum |W| + |b| per row), so
    the engine selects the bank kernel whose 24 inner sines go to v_sin_f32 unreduced (NWS_EXCITER_BANK_NOFRACT).  Same values as
    the v_fract form to rounding; the end-to-end exact vectors (test_e2e_*) run this variant against the reference."""
    import ctypes as C
    import nws_amd
    _lib = nws_amd._lib
    m, _ = models
    eng = m._engine
    w, _, dev = eng.weights()
    assert eng.bank_nofract_safe() and (w.exciter_opts & _lib.EXCITER_BANK_NOFRACT)
    w2 = _lib.NwsWeights.from_buffer_copy(w)
    w2.exciter_opts = w.exciter_opts & ~_lib.EXCITER_BANK_NOFRACT
    g = torch.Generator().manual_seed(78)
    B, T = 3, 21
    f0 = (120 + 500 * torch.rand(B, 1, T, generator=g)).cuda()
    control = torch.randn(B, 2, T, generator=g).cuda()
    pu = torch.rand(101, generator=g).cuda()
    _, film, _, _ = eng.frame_mlps(eng.control_gru(control))
    carry = eng.phase_carry(f0=f0[:, 0].contiguous())
    _, nf = eng.exciter_newt(f0[:, 0].contiguous(), None, carry, pu, film)
    fr = torch.empty_like(nf)
    _lib.check(_lib.lib().nws_exciter_newt(C.byref(w2), _lib.ptr(f0[:, 0].contiguous()), None, _lib.ptr(carry), _lib.ptr(pu),
                                           _lib.ptr(eng.rand_phase()), _lib.ptr(film), B, T, float(m.sample_rate), None,
                                           _lib.ptr(fr), _lib.stream_ptr()), "nws_exciter_newt (v_fract form)")
    torch.cuda.synchronize()
    d, scale = float((nf - fr).abs().max()), float(fr.abs().max())
    record("exact_bank_nofract_vs_fract", max_abs_diff=d, max_abs=scale)
    assert scale > 0 and d <= 1e-6 * max(1.0, scale), (d, scale)


def test_exciter_optional_table_fallbacks(models):
    """Every derived table of NwsWeights is optional for a C-ABI caller; each NULL selects another code path of the fused
    kernel.  Against the engine's default (all tables present, LUT range 6 -> kModeLutPairsDiv6):
      mixer_frags NULL  -> the workgroup splits the mixer weights itself: identical bits;
      lut_pairs NULL    -> kModeLut (two gathers, lerp with two roundings where the hot path has one FMA): <= 1 ulp-class;
      LUT range != 6    -> kModeLutPairs vs kModeLut on the same table: identical bits (the pair table holds the
                           reference's own intermediate fl(upper - lower))."""
    import ctypes as C
    import nws_amd
    _lib = nws_amd._lib
    _, fast = models
    eng = fast._engine
    w, _, dev = eng.weights()
    assert w.mixer_frags and w.lut_pairs and w.lut
    g = torch.Generator().manual_seed(78)
    B, T = 2, 19
    f0 = (90 + 700 * torch.rand(B, 1, T, generator=g)).cuda()
    control = torch.randn(B, 2, T, generator=g).cuda()
    pu = torch.rand(101, generator=g).cuda()
    gru = eng.control_gru(control)
    _, film, _, _ = eng.frame_mlps(gru)
    carry = eng.phase_carry(f0=f0[:, 0].contiguous())
    f0c = f0[:, 0].contiguous()

    def run(wx):
        out = torch.empty(B, 128 * T, device="cuda")
        _lib.check(_lib.lib().nws_exciter_newt(C.byref(wx), _lib.ptr(f0c), None, _lib.ptr(carry), _lib.ptr(pu),
                                               _lib.ptr(eng.rand_phase()), _lib.ptr(film), B, T, float(fast.sample_rate), None,
                                               _lib.ptr(out), _lib.stream_ptr()), "nws_exciter_newt")
        torch.cuda.synchronize()
        return out

    ref = run(w)
    scale = float(ref.abs().max())
    w_nf = _lib.NwsWeights.from_buffer_copy(w)
    w_nf.mixer_frags = None
    assert torch.equal(run(w_nf), ref)
    w_np = _lib.NwsWeights.from_buffer_copy(w)
    w_np.lut_pairs = None
    d = float((run(w_np) - ref).abs().max())
    assert d <= 2e-6 * max(1.0, scale), (d, scale)
    # a table range other than 6 (same table, relabelled: only the index arithmetic changes)
    w_r = _lib.NwsWeights.from_buffer_copy(w)
    w_r.lut_min, w_r.lut_max = -3.5, 3.5
    a = run(w_r)
    w_r2 = _lib.NwsWeights.from_buffer_copy(w_r)
    w_r2.lut_pairs = None
    assert torch.equal(a, run(w_r2))
    assert not torch.equal(a, ref)
    record("exciter_optional_tables", lut_two_gathers_vs_pairs_fma_max_abs=d, max_abs=scale)


def test_forward_audio_validates_row_blocks(models):
    """row blocks that do not tile the batch with even sizes (or come with the single-call event hooks) raise instead of
    leaving rows of the output unwritten."""
    _, m = models
    eng = m._engine
    B, T = 8, 4
    f0 = torch.full((B, 1, T), 220.0).cuda()
    c = torch.zeros(B, 2, T).cuda()
    pu, nz = torch.rand(101).cuda(), torch.rand(128 * T - 1).cuda()
    ws = eng.workspace(B, T) if hasattr(eng, "workspace") else None
    if ws is None:
        import ctypes as C
        from nws_amd import _lib
        plan = eng.reverb_aux(128 * T)[0]
        ws = torch.empty(_lib.lib().nws_forward_workspace_bytes(C.byref(plan), B, T), dtype=torch.uint8, device="cuda")
    eng.forward_control(f0, c, ws)
    ref = eng.forward_audio(f0, B, T, pu, nz, ws)
    ok = eng.forward_audio(f0, B, T, pu, nz, ws, row_blocks=[(0, 4), (4, 4)])
    assert torch.equal(ok, ref)
    for bad in ([(0, 4), (4, 2)], [(0, 3), (3, 5)], [(4, 4), (0, 4)], [(0, 4), (2, 6)]):
        with pytest.raises(RuntimeError, match="row_blocks"):
            eng.forward_audio(f0, B, T, pu, nz, ws, row_blocks=bad)
    with pytest.raises(RuntimeError, match="row_blocks"):
        eng.forward_audio(f0, B, T, pu, nz, ws, row_blocks=[(0, 4), (4, 4)], wait_event=torch.cuda.Event())


@pytest.mark.parametrize("inst", ["vn", "fl", "tpt"])
def test_range_proven_lookups_equal_the_clamped_form_bit_for_bit(inst, monkeypatch):
    """Round 5: where the staged FiLM rows and the worst-case exciter bound (NwsWeights.exciter_bound) prove that a tile's table
    indices stay inside the table, the fused tail runs floor-free lookups (v_fract / v_cvt_u32).  They must be the SAME bits as
    the clamped form of shaping.py:136-151 on every input: realistic F0, the timing script's torch.rand inputs, and FiLM
    parameters blown up so that a good part of the indices leave the table (the proof then fails and the clamped form runs)."""
    import nws_amd as nws

    nws.ensure_default_config()
    m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(ROOT, "tests", "golden", f"weights_{inst}.npz")).cuda().eval()
    m.newt = nws.FastNEWT(m.newt)
    g = torch.Generator(device="cuda").manual_seed(11)
    B, T = 6, 120
    cases = [(torch.rand(B, 1, T, device="cuda", generator=g), torch.rand(B, 2, T, device="cuda", generator=g)),
             (100 + 800 * torch.rand(B, 1, T, device="cuda", generator=g), torch.randn(B, 2, T, device="cuda", generator=g)),
             (100 + 800 * torch.rand(B, 1, T, device="cuda", generator=g), 6.0 * torch.randn(B, 2, T, device="cuda", generator=g))]
    pu, nz = torch.rand(101, device="cuda", generator=g), torch.rand(128 * T - 1, device="cuda", generator=g)
    with torch.no_grad():
        proven = [m(f0, c, phase_u=pu, noise=nz).clone() for f0, c in cases]
        assert m._engine.weights()[0].exciter_bound            # the bound table is in place
        monkeypatch.setenv("NWS_EXCITER_NO_RANGE", "1")
        m.invalidate_cache()
        assert not m._engine.weights()[0].exciter_bound
        clamped = [m(f0, c, phase_u=pu, noise=nz).clone() for f0, c in cases]
    for a, b in zip(proven, clamped):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_film_rows_as_fragment_records_by_lds_dma_equal_the_product_kernel():
    """Round 6 experiment kept alive (kOptFilmDma, profiles/r06/film_dma_ab.txt: measured as nothing, not on the product path): the
    oscillator kernel fed per-frame bf16x3 fragment records by LDS-DMA, F.upsample as (1 - w) p[f] + w p[f + 1] on the matrix pipe,
    per-frame range-proof masks - against the product kernel on the same inputs (odd T: a dead second hop; both utterance edges)."""
    import ctypes as C
    import nws_amd as nws
    _lib = nws._lib
    m = build_model(True)
    eng = m._engine
    w, _, _ = eng.weights()
    for B, T in ((3, 37), (2, 500)):
        g = torch.Generator(device="cuda").manual_seed(B)
        f0 = (80 + 900 * torch.rand(B, T, device="cuda", generator=g)).contiguous()
        control = torch.randn(B, 2, T, device="cuda", generator=g)
        carry = eng.phase_carry(f0=f0)
        _, film, _, _ = eng.frame_mlps(eng.control_gru(control))
        pu = torch.rand(101, device="cuda", generator=g)
        frags = torch.empty(B * T * (1536 + 16), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().nws_debug_film_frags(C.byref(w), film.data_ptr(), B, T, frags.data_ptr(), _lib.stream_ptr()))
        outs = {}
        for v, src in ((44, film), (108, frags)):
            outs[v] = torch.zeros(B, 128 * T, device="cuda")
            _lib.check(_lib.lib().nws_debug_exciter_newt(v, C.byref(w), f0.data_ptr(), carry.data_ptr(), pu.data_ptr(), eng.rand_phase().data_ptr(),
                                                         src.data_ptr(), B, T, 16000.0, outs[v].data_ptr(), _lib.stream_ptr()))
        _, ref = eng.exciter_newt(f0, None, carry, pu, film)
        assert torch.equal(outs[44], ref)                       # variant 44 IS the product kernel
        d = float((outs[108] - ref).abs().max())
        record(f"film_dma_variant_B{B}_T{T}", maxabs=d, signal_rms=float(ref.pow(2).mean().sqrt()))
        assert d <= 2e-7, d
