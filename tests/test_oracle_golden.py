"""Pin the oracle (oracle/newt_oracle.py) against vectors recorded from the REAL reference.

The reference ships no tests; tests/golden/make_golden.py ran the reference
itself in the build container and these vectors are what it produced.
"""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_npz, rms
from oracle.newt_oracle import OracleNEWT


@pytest.fixture(scope="module")
def oracles(weights):
    return (OracleNEWT(weights, fast=False), OracleNEWT(weights, fast=True, lut_python_loop=False))


def _check_e2e(oracles, g, draws, tol=2e-6):
    exact, fast = oracles
    y = exact(g["f0"], g["control"], draws["phase_u"], draws["noise"]).numpy()
    yf = fast(g["f0"], g["control"], draws["phase_u"], draws["noise"]).numpy()
    assert rms(y - g["y_newt"]) <= tol, rms(y - g["y_newt"])
    assert rms(yf - g["y_fast"]) <= tol, rms(yf - g["y_fast"])
    # exact NEWT and FastNEWT are different functions: keep comparing like with like
    assert rms(g["y_newt"] - g["y_fast"]) > 1e-6


def test_g1_realistic(oracles, g1):
    _check_e2e(oracles, g1, g1)


def test_g2_rand_inputs(oracles, g1):
    _check_e2e(oracles, load_npz("g2_rand.npz"), g1)   # same seed -> same draws as g1


def test_g6_high_f0(oracles, g1):
    _check_e2e(oracles, load_npz("g6_highf0.npz"), g1, tol=5e-6)


@pytest.mark.parametrize("inst", ["fl", "tpt"])
def test_g7_other_instruments(inst):
    """The other two shipped checkpoints (flute, trumpet): their input_scale ranges push the exact shapers' sine arguments
    further than the violin's; oracle pinned on a realistic 1 s vector each."""
    w = {k: v for k, v in load_npz(f"weights_{inst}.npz").items() if not k.startswith("__")}
    g = load_npz(f"g7_{inst}.npz")
    _check_e2e((OracleNEWT(w, fast=False), OracleNEWT(w, fast=True, lut_python_loop=False)), g, g)


def test_g4_streaming_sizes(oracles):
    g = load_npz("g4_stream.npz")
    exact, fast = oracles
    for T in (2, 32):
        y = exact(g[f"f0_T{T}"], g[f"control_T{T}"], g[f"phase_u_T{T}"], g[f"noise_T{T}"]).numpy()
        yf = fast(g[f"f0_T{T}"], g[f"control_T{T}"], g[f"phase_u_T{T}"], g[f"noise_T{T}"]).numpy()
        assert y.shape == (2, 128 * T)
        assert rms(y - g[f"y_newt_T{T}"]) <= 2e-6
        assert rms(yf - g[f"y_fast_T{T}"]) <= 2e-6


def test_g3_every_stage(oracles):
    g = load_npz("g3_stages.npz")
    exact, fast = oracles
    st, stf = {}, {}
    exact(g["f0"], g["control"], g["phase_u"], g["noise"], stages=st)
    fast(g["f0"], g["control"], g["phase_u"], g["noise"], stages=stf)
    # bit-exact stages
    for k in ("f0_up", "osc"):
        assert np.array_equal(st[k].numpy(), g[k]), k
    tight = dict(exciter=1e-6, gru_out=1e-6, embedding=1e-6, film=2e-6, lut_arg=2e-6, H=2e-6, noise_out=1e-7)
    for k, tol in tight.items():
        assert np.max(np.abs(st[k].numpy() - g[k])) <= tol, (k, np.max(np.abs(st[k].numpy() - g[k])))
    assert np.max(np.abs(st["shaped"].numpy() - g["shaped_exact"])) <= 2e-5
    assert np.max(np.abs(stf["shaped"].numpy() - g["shaped_lut"])) <= 2e-5
    assert np.max(np.abs(st["newt_out"].numpy() - g["newt_out_exact"])) <= 1e-6
    assert np.max(np.abs(stf["newt_out"].numpy() - g["newt_out_lut"])) <= 1e-6
    assert rms(st["y"].numpy() - g["y_exact"]) <= 2e-6
    assert rms(stf["y"].numpy() - g["y_lut"]) <= 2e-6
    # extra control channels are ignored (reference quirk)
    y2 = exact(g["f0"], g["control"][:, :2], g["phase_u"], g["noise"]).numpy()
    assert np.array_equal(y2, st["y"].numpy())


def test_g5_lut_bit_exact(oracles):
    g = load_npz("g5_lut.npz")
    exact, fast = oracles
    table = fast.lookup_table()
    assert hashlib.sha256(table.numpy().tobytes()).digest() == g["table_sha256"].tobytes()
    assert np.array_equal(table[g["rows"]].numpy(), g["table_rows"])
    xp = torch.from_numpy(g["probes"]).view(1, 1, -1).expand(1, 64, -1).contiguous()
    assert np.array_equal(fast.lut_shaper(xp)[0].numpy(), g["probe_out"])
    # python-loop lookup (the reference's form) == gather
    loop = OracleNEWT(fast.w, fast=True, lut_python_loop=True)
    assert np.array_equal(loop.lut_shaper(xp)[0].numpy(), g["probe_out"])
    # below-range extrapolates linearly, above-range is flat
    assert np.array_equal(g["probe_out"][:, -1], g["probe_out"][:, -2])


@pytest.mark.parametrize("name", ["g8_small", "g8_odd", "g8_oddlen"])
def test_g8_non_default_gin_configurations(name):
    """The oracle is generic in every gin-configurable size; pinned on two non-default configurations recorded from the
    real reference (random init; 60 harmonics / 32 shapers of width 16, depth 3 / GRU 96 / embedding 80 / hop 64 / 128-tap
    FIR / two NEWT output channels / 1 s reverb, and an odd one: 7 / 5 / width 3, depth 2 / 33 / 17 / hop 10 / 30 taps / 8 kHz)."""
    import json

    z = load_npz(name + ".npz")
    w = {k: v for k, v in z.items() if not k.startswith("__")}
    hp = json.loads(str(z["__hparams__"]))
    kw = dict(sample_rate=hp["sample_rate"], control_hop=hp["control_hop"], table_size=int(z["__table_size__"]),
              table_min=float(z["__table_min__"]), table_max=float(z["__table_max__"]))
    exact, fast = OracleNEWT(w, fast=False, **kw), OracleNEWT(w, fast=True, lut_python_loop=False, **kw)
    st = {}
    y = exact(z["__f0__"], z["__control__"], z["__phase_u__"], z["__noise__"], stages=st).numpy()
    yf = fast(z["__f0__"], z["__control__"], z["__phase_u__"], z["__noise__"]).numpy()
    assert np.array_equal(st["osc"].numpy(), z["__osc__"])                   # bit-exact oscillator bank
    assert np.array_equal(fast.lookup_table().numpy(), z["__lookup_table__"])
    for k, tol in dict(exciter=1e-6, embedding=1e-6, film=2e-6, H=2e-6).items():
        assert np.max(np.abs(st[k].numpy() - z[f"__{k}__"])) <= tol, k
    assert np.max(np.abs(st["noise_out"].numpy() - z["__noise_out__"][:, 0])) <= 1e-6
    assert np.max(np.abs(st["pre_reverb"].numpy() - z["__pre_reverb__"])) <= 1e-5
    assert y.shape == z["__y_newt__"].shape
    assert rms(y - z["__y_newt__"]) <= 2e-6, rms(y - z["__y_newt__"])
    assert rms(yf - z["__y_fast__"]) <= 2e-6, rms(yf - z["__y_fast__"])


def test_rng_draw_order_matches_reference(oracles, g1):
    """Drawing inside the oracle consumes torch's CPU generator in the reference's order and sizes."""
    exact, _ = oracles
    torch.manual_seed(1234)
    y = exact(g1["f0"], g1["control"]).numpy()
    assert rms(y - g1["y_newt"]) <= 2e-6


def test_g10_timing_script_model():
    """BASELINE config 1 literally (scripts/time_forward_pass.py:27-43): UNMODIFIED random-init NeuralWaveshaping() under
    gin/models/newt.gin, torch.rand inputs at T = 500, both shapers - recorded from the real reference with fixed seeds."""
    z = load_npz("g10_timing_script.npz")
    w = {k: v for k, v in z.items() if not k.startswith("__")}
    g = {"f0": z["__f0__"], "control": z["__control__"], "phase_u": z["__phase_u__"], "noise": z["__noise__"],
         "y_newt": z["__y_newt__"], "y_fast": z["__y_fast__"]}
    assert g["f0"].shape == (1, 1, 500) and float(g["f0"].max()) < 1.0 and g["y_newt"].shape == (1, 64000)
    _check_e2e((OracleNEWT(w, fast=False), OracleNEWT(w, fast=True, lut_python_loop=False)), g, g, tol=5e-6)
