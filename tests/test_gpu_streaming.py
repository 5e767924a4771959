"""-m gpu tests of the stateful streaming path.  Anchor: the concatenation of the streamed chunks must equal
the oracle's ONE-SHOT forward up to the reverb input (stage `pre_reverb`, recorded semantics of the reference), for any
chunking; the streaming reverb is a linear convolution, checked against a float64 convolution of that same signal."""
import numpy as np
import pytest
import torch
from scipy.signal import fftconvolve

from conftest import rms
from gpu_util import build_model, maxabs, record

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(weights):
    from oracle.newt_oracle import OracleNEWT

    return build_model(True), OracleNEWT(weights, fast=True, lut_python_loop=False), weights


@pytest.mark.parametrize("chunks,B", [([60], 3), ([1, 7, 16, 4, 31, 1], 3), ([2] * 30, 3), ([13, 47], 3),
                                      ([1, 7, 16, 4, 31, 1], 17), ([2] * 12 + [20], 33)])   # >= 16 streams: the shared-noise MFMA kernel on windows
def test_stream_equals_one_shot(setup, chunks, B):
    model, oracle, weights = setup
    F = sum(chunks)
    g = torch.Generator().manual_seed(F * 7 + len(chunks))
    f0 = (120 + 600 * torch.rand(B, 1, 1, generator=g)) * (1 + 0.03 * torch.randn(B, 1, F, generator=g))
    control = torch.randn(B, 2, F, generator=g)
    pu, nz = torch.rand(101, generator=g), torch.rand(128 * F - 1, generator=g)
    st = {}
    oracle(f0, control, pu, nz, stages=st)
    pre_ref = st["pre_reverb"].numpy()
    s = model.stream(B, phase_u=pu.cuda(), noise=nz.cuda())
    ys, pres, k = [], [], 0
    for i, K in enumerate(chunks):
        y = s.push(f0[:, :, k:k + K].cuda(), control[:, :, k:k + K].cuda(), final=(i == len(chunks) - 1))
        ys.append(y.cpu().numpy())
        pres.append(s._last_pre.cpu().numpy())
        k += K
    pre = np.concatenate(pres, axis=1)
    y = np.concatenate(ys, axis=1)
    assert pre.shape == (B, 128 * F) and y.shape == (B, 128 * F) and s.samples_emitted == 128 * F
    e_pre = maxabs(pre, pre_ref)
    ir_ = np.concatenate([[0.0], weights["reverb.ir"][0].astype(np.float64)])
    full = np.stack([fftconvolve(pre_ref[b].astype(np.float64), ir_) for b in range(B)])
    y_ref = pre_ref + full[:, :128 * F]
    e_y = rms(y - y_ref)
    tail_ref = full[:, 128 * F:128 * F + 32000]
    tail = s.reverb_tail().cpu().numpy()[:, :tail_ref.shape[1]]
    e_tail = rms(tail - tail_ref)
    record(f"stream_chunks_{len(chunks)}x_B{B}", pre_max_abs_err=e_pre, pre_max=float(np.abs(pre_ref).max()), y_rms_err=e_y,
           y_rms=rms(y_ref), tail_rms_err=e_tail, tail_rms=rms(tail_ref))
    assert e_pre <= 2e-6 * max(1.0, float(np.abs(pre_ref).max()) / 1e-2)   # pre-reverb level is ~1e-2: ~1e-6 absolute
    assert e_y <= 1e-4 and e_tail <= 1e-4


def test_stream_self_drawn_noise_and_long_chunks(setup):
    model, _, _ = setup
    B, F = 2, 300                       # one 300-frame chunk is split internally (linear-reverb chunk limit 249 frames)
    f0 = 200 + 50 * torch.rand(B, 1, F, device="cuda")
    control = torch.randn(B, 2, F, device="cuda")
    torch.manual_seed(3)
    s1 = model.stream(B)
    a = s1.push(f0, control, final=True)
    torch.manual_seed(3)
    s2 = model.stream(B)
    b = torch.cat([s2.push(f0[:, :, :100], control[:, :, :100]), s2.push(f0[:, :, 100:], control[:, :, 100:], final=True)], 1)
    assert a.shape == (B, 128 * F) and b.shape == (B, 128 * F) and torch.isfinite(a).all()
    # NEWT branch and state handling are chunking-independent; the self-drawn noise differs with the chunking of the draws,
    # so only the level is compared here (the noise branch is ~-60 dB)
    assert abs(float(a.std()) - float(b.std())) <= 0.05 * float(a.std())
    with pytest.raises(RuntimeError):
        s1.push(f0[:, :, :4], control[:, :, :4])


def test_captured_hop_with_static_io_equals_push(setup):
    """Steady-state hops replay a hipGraph; `static_io` / `hop` expose the captured hop's own buffers (no copies).  Same
    stream contents as eager pushes, bit for bit (same kernels, same launch arguments), for injected and for drawn noise."""
    model, _, _ = setup
    B, K, n = 2, 2, 12
    g = torch.Generator().manual_seed(9)
    f0 = (150 + 300 * torch.rand(B, 1, 1, generator=g)) * (1 + 0.02 * torch.randn(B, 1, K * n, generator=g))
    control = torch.randn(B, 2, K * n, generator=g)
    pu, nz = torch.rand(101, generator=g), torch.rand(128 * K * n - 1, generator=g)
    chunks = [(f0[:, :, i * K:(i + 1) * K].cuda(), control[:, :, i * K:(i + 1) * K].cuda()) for i in range(n)]
    eager = model.stream(B, phase_u=pu.cuda(), noise=nz.cuda(), graph=False)
    ref = [eager.push(a, c, final=(i == n - 1)) for i, (a, c) in enumerate(chunks)]
    s = model.stream(B, phase_u=pu.cuda(), noise=nz.cuda())
    got = []
    for i, (a, c) in enumerate(chunks):
        if 4 <= i < n - 1:
            f0_in, c_in, out = s.static_io(K)
            f0_in.copy_(a[:, 0])
            c_in.copy_(c)
            got.append(s.hop(K).clone())
        else:
            got.append(s.push(a, c, final=(i == n - 1)))
    assert s._graphs, "the steady-state hop was never captured"
    for i, (x, y) in enumerate(zip(got, ref)):
        assert torch.equal(x, y), i
    assert s.samples_emitted == eager.samples_emitted == 128 * K * n
    # drawn noise: the graph carries the draws; same seed -> same stream as eager pushes drawing chunk by chunk
    outs = []
    for use_graph in (False, True):
        torch.manual_seed(11)
        st = model.stream(B, graph=use_graph)
        outs.append(torch.cat([st.push(a, c) for a, c in chunks[:8]], dim=1))
    assert torch.isfinite(outs[1]).all() and abs(float(outs[0].std()) - float(outs[1].std())) <= 0.05 * float(outs[0].std())


def test_stream_picks_up_weight_updates_and_recaptures(setup):
    """a stream must not keep convolving with a stale IR (or replay a graph that points into freed tables) after the
    model's weights change.  In-place update of reverb.ir in the middle of a stream: the hops after the engine has noticed it
    (the next forward, or the stream's own periodic fingerprint walk) equal those of a fresh stream run on the new weights from
    the same state - checked through the pre-reverb tap (unchanged by the IR) and the output (changed)."""
    import copy

    model, _, _ = setup
    m = copy.deepcopy(model)
    B, K = 1, 2
    g = torch.Generator().manual_seed(21)
    f0 = (200 + 100 * torch.rand(B, 1, K * 12, generator=g)).cuda()
    control = torch.randn(B, 2, K * 12, generator=g).cuda()
    pu, nz = torch.rand(101, generator=g).cuda(), torch.rand(128 * K * 12 - 1, generator=g).cuda()
    s = m.stream(B, phase_u=pu, noise=nz)
    outs = [s.push(f0[:, :, i * K:(i + 1) * K], control[:, :, i * K:(i + 1) * K]) for i in range(6)]
    assert s._graphs                                             # steady state reached, hop captured
    with torch.no_grad():
        m.reverb.ir.mul_(0.0)                                    # in place: same storage, version bumped
    m(f0[:, :, :4].contiguous(), control[:, :, :4].contiguous())   # any forward makes the engine rebuild its tables
    y = s.push(f0[:, :, 12:14], control[:, :, 12:14])
    # with a zero IR the output is the dry pre-reverb signal itself
    assert torch.equal(y, s._last_pre) and not torch.equal(outs[-1], outs[-2])
    assert not s._graphs or s._w_seen is m._engine._w


def test_stream_geometry_follows_the_reverb_length():
    """the stream's chunk limit and ring come from THIS model's reverb, not from the 16 kHz default.  An 8 kHz model
    (2 s reverb = 15999 taps, FFT plan of 32000 points) takes chunks of at most (32000 - 15999) / 128 - 1 = 124 frames - one such
    chunk equals the same frames pushed in pieces; a 22.05 kHz model (44099 taps) does not fit the 32767 samples of reverb history
    a stream keeps and is refused at construction with the reason, instead of failing every push with NWS_ERR_BAD_ARG."""
    import nws_amd as nws
    nws.ensure_default_config()
    try:
        nws.gin.parse_config("NeuralWaveshaping.sample_rate = 8000\nHarmonicOscillator.sample_rate = 8000\nReverb.sr = 8000\n")
        torch.manual_seed(11)
        m = nws.NeuralWaveshaping().cuda().eval()
        m.newt = nws.FastNEWT(m.newt)
        assert m.reverb.ir.numel() == 15999 and m._engine.specialised()
        g = torch.Generator().manual_seed(3)
        F = 124
        f0 = (150 + 300 * torch.rand(2, 1, 1, generator=g)) * torch.ones(2, 1, F)
        control = torch.randn(2, 2, F, generator=g)
        pu, nz = torch.rand(101, generator=g).cuda(), torch.rand(128 * F - 1, generator=g).cuda()
        outs = []
        for chunks in ([124], [60, 64], [3] * 40 + [4]):
            s = m.stream(2, phase_u=pu, noise=nz)
            assert s.max_frames == 124 and s._plan_n == 32000
            k, ys = 0, []
            for i, K in enumerate(chunks):
                ys.append(s.push(f0[:, :, k:k + K].cuda(), control[:, :, k:k + K].cuda(), final=(i == len(chunks) - 1)))
                k += K
            outs.append(torch.cat(ys, dim=1).cpu().numpy())
        assert outs[0].shape == (2, 128 * F) and np.isfinite(outs[0]).all() and rms(outs[0]) > 1e-5
        for o in outs[1:]:
            assert rms(o - outs[0]) <= 2e-6 * max(rms(outs[0]), 1e-3), rms(o - outs[0])
        nws.gin.parse_config("NeuralWaveshaping.sample_rate = 22050\nHarmonicOscillator.sample_rate = 22050\nReverb.sr = 22050\n")
        m22 = nws.NeuralWaveshaping().cuda().eval()
        assert m22.reverb.ir.numel() == 44099
        with pytest.raises(RuntimeError, match="reverb history"):
            m22.stream(1)
        y = m22(torch.full((1, 1, 8), 200.0).cuda(), torch.randn(1, 2, 8).cuda())      # the one-shot forward serves it
        assert y.shape == (1, 1024) and torch.isfinite(y).all()
    finally:
        nws.gin.clear_config()
        nws.gin.parse_config_file(nws.DEFAULT_GIN)


def test_stream_refresh_picks_up_a_weight_update_at_once(setup):
    """a captured hop replays the old tables for up to 250 ms after an in-place weight update; refresh() closes the
    window - the very next hop of a graph=True stream equals an eager stream's."""
    import copy
    model = copy.deepcopy(setup[0])
    g = torch.Generator().manual_seed(5)
    K = 2
    pu = torch.rand(101, generator=g).cuda()
    nz = torch.rand(128 * 40, generator=g).cuda()
    f0 = (200 + 100 * torch.rand(1, 1, 40, generator=g)).cuda()
    c = torch.randn(1, 2, 40, generator=g).cuda()
    a, b = model.stream(1, phase_u=pu, noise=nz, graph=True), model.stream(1, phase_u=pu, noise=nz, graph=False)
    for i in range(6):
        ya, yb = a.push(f0[:, :, 2 * i:2 * i + K], c[:, :, 2 * i:2 * i + K]), b.push(f0[:, :, 2 * i:2 * i + K], c[:, :, 2 * i:2 * i + K])
    assert a._graphs and torch.equal(ya, yb)
    with torch.no_grad():
        model.newt.mixer[0].weight.mul_(0.5)           # in-place: nobody tells the engine
    a.refresh()
    b.refresh()
    assert not a._graphs                                   # dropped, to be re-captured on the new tables
    ya, yb = a.push(f0[:, :, 12:14], c[:, :, 12:14]), b.push(f0[:, :, 12:14], c[:, :, 12:14])
    assert torch.equal(ya, yb)


def test_fused_hop_is_bit_identical_with_the_seven_launch_form(tmp_path):
    """Hops of <= 256 samples run in four launches (csrc/stream.hip): the reverb's history parts, the per-utterance head and the
    frame MLPs of the new frames as extra workgroups of the recurrence launch, reverb part 0 + hand-over inside the closing
    kernel.  Same code, same order of sums: the emitted samples of 40 graph-replayed hops (final one included) must equal the
    seven-launch form's (NWS_STREAM_SPLIT_REVERB=0) bit for bit.  The switches are read once per process."""
    import os
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "stream_hop_ab.py")
    outs = {}
    # launch structures with the SAME frame-MLP arithmetic must agree bit for bit: seven / five launches with the 32-frame tile
    # kernel; five launches / the default four with the matrix-vector form of csrc/mlp_few.h for the hop's two frames (exact fp32
    # products of the same 22-bit weights: equal to the tile kernel's results to fp32 rounding)
    variants = {"seven_tiles": {"NWS_STREAM_SPLIT_REVERB": "0", "NWS_MLP_FEW": "0"}, "five_tiles": {"NWS_MLP_FEW": "0"},
                "five_few": {"NWS_STREAM_FUSE_MLP": "0"}, "default": {}}
    for name, env in variants.items():
        path = str(tmp_path / f"hop_{name}.npy")
        r = subprocess.run([sys.executable, tool, path, "3", "dump-only"], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(path)
    a = outs["seven_tiles"]
    assert a.shape == (3, 128 * 80) and np.isfinite(a).all() and rms(a) > 1e-3
    assert np.array_equal(a, outs["five_tiles"])
    assert np.array_equal(outs["five_few"], outs["default"])
    assert rms(outs["default"] - a) <= 1e-5 * rms(a), rms(outs["default"] - a)      # measured 3.4e-6 (1e-7 absolute; the parity bar is 1e-4)
