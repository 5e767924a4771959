"""-m gpu tests of the sub-modules called on their own (the module classes are public interface) against the
stage taps recorded from the real reference (tests/golden/g3_stages.npz, B=2, T=3), through torch.ops.newt_hip.* - and, in a
second process, through the ctypes binding of the same C-ABI (NWS_BACKEND=ctypes)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_npz, rms
from gpu_util import build_model, dev, maxabs, record

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    return load_npz("g3_stages.npz")


@pytest.fixture(scope="module")
def models():
    return build_model(False), build_model(True)


def test_backend_is_the_op_layer_unless_ctypes_was_asked_for():
    import nws_amd  # noqa: F401
    from nws_amd import engine
    o = engine.ops()
    if os.environ.get("NWS_BACKEND") == "ctypes":
        assert o is None
    else:
        assert o is torch.ops.newt_hip and int(o.abi_version()) == 6
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            o.sine(torch.zeros(4))
        with pytest.raises(RuntimeError, match="multiple of 128"):
            o.oscillator(torch.zeros(2, 100, device="cuda"), torch.zeros(101, device="cuda"), torch.zeros(101, device="cuda"), 16000.0)


def test_harmonic_oscillator_forward(models, g):
    exact, _ = models
    osc = exact.osc(dev(g["f0_up"]), phase_u=dev(g["phase_u"]))
    e = maxabs(osc.cpu().numpy(), g["osc"])
    record("module_osc", max_abs_err=e)
    assert osc.shape == (2, 101, 384) and e <= 2e-6, e
    # the anti-alias mask is exact: zeros where the reference has zeros
    assert np.array_equal(osc.cpu().numpy() == 0.0, g["osc"] == 0.0)
    # default path draws like the reference (rand_like on the module's device) and is seed-reproducible
    torch.manual_seed(3)
    a = exact.osc(dev(g["f0_up"]))
    torch.manual_seed(3)
    b = exact.osc(dev(g["f0_up"]))
    assert torch.equal(a, b) and not torch.equal(a, osc)
    # a length that is not a multiple of 128 takes the runtime-size kernels (csrc/generic.hip): same values
    c = exact.osc(dev(g["f0_up"])[:, :100].contiguous(), phase_u=dev(g["phase_u"]))
    assert c.shape == (2, 101, 100) and maxabs(c.cpu().numpy(), g["osc"][:, :, :100]) <= 2e-6
    with pytest.raises(RuntimeError):
        exact.osc(torch.zeros(1, 2, 100, device="cuda"))       # (B, N) expected


def test_control_module_and_time_distributed_mlps(models, g):
    exact, _ = models
    control = dev(g["control"])
    emb = exact.embedding(control[:, :2])
    e = maxabs(emb.cpu().numpy(), g["embedding"])
    assert emb.shape == (2, 128, 3) and e <= 2e-5, e
    film = exact.newt.mlp(dev(g["embedding"]))
    ef = maxabs(film.cpu().numpy(), g["film"])
    H = exact.h_generator(dev(g["embedding"]))
    eh = maxabs(H.cpu().numpy(), g["H"])
    record("module_mlps", emb_max_abs_err=e, film_max_abs_err=ef, H_max_abs_err=eh)
    assert film.shape == (2, 256, 3) and ef <= 2e-5 * max(1.0, float(np.abs(g["film"]).max())), ef
    assert H.shape == (2, 129, 3) and eh <= 2e-5 * max(1.0, float(np.abs(g["H"]).max())), eh
    with pytest.raises(RuntimeError):
        exact.h_generator(torch.zeros(1, 64, 3, device="cuda"))      # wrong channel count


def test_layer_norm_film_sine(models):
    exact, _ = models
    gcpu = torch.Generator().manual_seed(1)
    x = torch.randn(3, 128, 7, generator=gcpu)
    ln = exact.newt.mlp.net[1]
    ref = torch.nn.functional.layer_norm(x.transpose(1, 2), (128,), ln.layer_norm.weight.detach().cpu(),
                                         ln.layer_norm.bias.detach().cpu(), ln.layer_norm.eps).transpose(1, 2)
    assert maxabs(ln(x.cuda()).cpu().numpy(), ref.numpy()) <= 2e-6
    film = exact.newt.waveshaping_index
    a, gm, bt = torch.randn(2, 64, 50, generator=gcpu), torch.randn(2, 64, 50, generator=gcpu), torch.randn(2, 64, 1, generator=gcpu)
    assert torch.equal(film(a.cuda(), gm.cuda(), bt.cuda()).cpu(), gm * a + bt)      # multiply then add, two roundings: bit-exact
    from nws_amd import Sine
    v = (torch.rand(1000, generator=gcpu) - 0.5) * 200
    assert maxabs(Sine()(v.cuda()).cpu().numpy(), np.sin(v.double().numpy())) <= 3e-7


def test_newt_and_fastnewt_forward(models, g):
    exact, fast = models
    exc, emb = dev(g["exciter"]), dev(g["embedding"])
    out = exact.newt(exc, emb)
    outf = fast.newt(exc, emb)
    e, ef = maxabs(out[:, 0].cpu().numpy(), g["newt_out_exact"]), maxabs(outf[:, 0].cpu().numpy(), g["newt_out_lut"])
    record("module_newt", exact_max_abs_err=e, lut_max_abs_err=ef, out_max=float(np.abs(g["newt_out_exact"]).max()))
    assert out.shape == (2, 1, 384) and e <= 2e-6, e
    assert outf.shape == (2, 1, 384) and ef <= 2e-6, ef
    with pytest.raises(RuntimeError):
        exact.newt(exc[:, :, :256].contiguous(), emb)              # 256 samples against 3 frames


def test_fir_noise_synth_forward(models, g):
    exact, _ = models
    out = exact.noise_synth(dev(g["H"]), noise=dev(g["noise"]))
    e = maxabs(out[:, 0].cpu().numpy(), g["noise_out"])
    record("module_noise", max_abs_err=e, out_max=float(np.abs(g["noise_out"]).max()))
    assert out.shape == (2, 1, 384) and e <= 1e-5 * max(1.0, float(np.abs(g["noise_out"]).max())), e
    torch.manual_seed(5)
    a = exact.noise_synth(dev(g["H"]))
    torch.manual_seed(5)
    assert torch.equal(a, exact.noise_synth(dev(g["H"])))


def test_forward_rebuilt_from_the_sub_modules(models, g):
    """The reference's forward (models/neural_waveshaping.py:74-90) spelled out with the sub-modules, one stage kernel per call,
    against the reference's own output and the fused forward."""
    exact, fast = models
    f0, control = dev(g["f0"]), dev(g["control"])
    pu, nz = dev(g["phase_u"]), dev(g["noise"])
    for model, key in ((exact, "y_exact"), (fast, "y_lut")):
        f0_up = torch.nn.functional.interpolate(f0, size=f0.shape[-1] * 128, mode="linear")       # torch plumbing, as in the reference
        osc = model.osc(f0_up[:, 0].contiguous(), phase_u=pu)
        exciter = torch.nn.functional.conv1d(osc, model.harmonic_mixer.weight, model.harmonic_mixer.bias)   # stock nn.Conv1d
        emb = model.embedding(control[:, :2])
        # the stock Conv1d's output carries a graph; the HIP stage kernels are inference-only and say so instead of silently
        # returning a graph-less tensor
        with pytest.raises(RuntimeError, match="inference-only"):
            model.newt(exciter.contiguous(), emb)
        with torch.no_grad():
            newt_out = model.newt(exciter.contiguous(), emb)
            noise_out = model.noise_synth(model.h_generator(emb), noise=nz)
            y = model.reverb((newt_out + noise_out)[:, 0].contiguous())
        e = rms(y.cpu().numpy() - g[key])
        fused = model(f0, control, phase_u=pu, noise=nz)
        record("module_chain_" + key, rms_err=e, rms_vs_fused=rms((y - fused).cpu().numpy()))
        assert e <= 1e-5, (key, e)


def test_in_place_parameter_updates_are_noticed(models, g):
    """the engine caches raw pointers and derived tables; in-place updates must not leave them stale."""
    import copy
    _, fast = models
    m = copy.deepcopy(fast)
    f0, control, pu, nz = dev(g["f0"]), dev(g["control"]), dev(g["phase_u"]), dev(g["noise"])
    y0 = m(f0, control, phase_u=pu, noise=nz)
    assert torch.equal(y0, fast(f0, control, phase_u=pu, noise=nz))          # the copy has its own, working engine
    with torch.no_grad():
        m.reverb.ir.mul_(0.5)                                                  # in place: same storage, version bumped
    y1 = m(f0, control, phase_u=pu, noise=nz)
    assert not torch.equal(y0, y1)
    with torch.no_grad():
        m.reverb.ir.mul_(2.0)
        m.harmonic_mixer.weight.add_(0.0)
    assert torch.equal(m(f0, control, phase_u=pu, noise=nz), y0)
    sd = {k: v.clone() for k, v in m.newt.state_dict().items()}
    sd["mixer.0.bias"] += 1.0
    m.newt.load_state_dict(sd)                                                 # sub-module load: in-place copy_
    assert not torch.equal(m(f0, control, phase_u=pu, noise=nz), y0)
    m.reverb.ir = torch.nn.Parameter(torch.zeros_like(m.reverb.ir))           # a replaced Parameter object
    y_dry = m(f0, control, phase_u=pu, noise=nz)
    assert torch.isfinite(y_dry).all() and not torch.equal(y_dry, y1)
    # mixed devices / CPU inputs raise like the reference
    with pytest.raises(RuntimeError):
        m(f0.cpu(), control, phase_u=pu, noise=nz)


@pytest.mark.skipif(os.environ.get("NWS_BACKEND") == "ctypes", reason="already the ctypes pass")
def test_same_suite_through_the_ctypes_binding():
    """The torch-free binding of the same C-ABI (what INTEGRATION.md shows a foreign host): module tests + end-to-end parity."""
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_modules.py"),
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_streaming.py"),
                        os.path.join(ROOT, "tests", "test_gpu_generic.py"),
                        "-k", "not offline_render and not soak and not full_size"],
                       env=dict(os.environ, NWS_BACKEND="ctypes"), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_forward_hooks_on_sub_modules_fire(models, g):
    """The reference's users (and tests/golden/make_golden.py itself) tap stages with forward hooks on sub-modules
    (models/neural_waveshaping.py:76-88 calls them one by one).  The fused forward never calls the sub-modules, so with hooks
    registered the forward runs module by module - the reference's own sequence, one HIP stage kernel each - and the taps are
    the reference's taps; without hooks the fused kernels run again."""
    exact, fast = models
    f0, control = dev(g["f0"]), dev(g["control"])
    pu, nz = dev(g["phase_u"]), dev(g["noise"])
    for model, ykey, shaped_key in ((exact, "y_exact", "shaped_exact"), (fast, "y_lut", "shaped_lut")):
        y_fused = model(f0, control, phase_u=pu, noise=nz)
        taps = {}

        def tap(name, what="out"):
            def hook(mod, inp, out):
                taps[name] = (inp[0] if what == "in" else out).detach().clone()
            return hook

        hs = [model.osc.register_forward_hook(tap("osc")), model.harmonic_mixer.register_forward_hook(tap("exciter")),
              model.embedding.register_forward_hook(tap("embedding")), model.newt.mlp.register_forward_hook(tap("film")),
              model.newt.waveshaping_index.register_forward_hook(tap("lut_arg")),
              model.newt.normalising_coeff.register_forward_hook(tap("shaped", "in")),
              model.h_generator.register_forward_hook(tap("H")), model.noise_synth.register_forward_hook(tap("noise_out")),
              model.reverb.register_forward_hook(tap("pre_reverb", "in"))]
        try:
            y = model(f0, control, phase_u=pu, noise=nz)
        finally:
            for h in hs:
                h.remove()
        assert set(taps) == {"osc", "exciter", "embedding", "film", "lut_arg", "shaped", "H", "noise_out", "pre_reverb"}
        for k, tol in dict(osc=2e-6, exciter=2e-5, embedding=1e-5, film=5e-5, lut_arg=5e-5, H=5e-5).items():
            assert maxabs(taps[k].cpu().numpy(), g[k]) <= tol, k
        assert maxabs(taps["shaped"].cpu().numpy(), g[shaped_key]) <= 5e-5
        assert maxabs(taps["noise_out"].cpu().numpy()[:, 0], g["noise_out"]) <= 1e-6
        e = rms(y.cpu().numpy() - g[ykey])
        record("hooked_forward_" + ykey, rms_err=e, rms_vs_fused=rms((y - y_fused).cpu().numpy()))
        assert e <= 1e-5
        assert torch.equal(model(f0, control, phase_u=pu, noise=nz), y_fused)       # hooks gone: the fused kernels again
