"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/nws_hip.h declares,
host-side planning functions, the gin reader, checkpoint loading, the module surface, loud failure without a GPU."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT

nws = importlib.import_module("neural-waveshaping-synthesis_amd")
_lib = importlib.import_module("neural-waveshaping-synthesis_amd._lib")
REF_GIN = "/root/reference/gin/models/newt.gin"


def _declared_symbols(headers=("nws_hip.h", "nws_hip_debug.h")):
    found = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        found |= set(re.findall(r"\b(nws_[a-z0-9_]+)\s*\(", text))
    return sorted(found)


def test_product_header_holds_no_diagnostic_entry_points():
    """Timing ablations live in include/nws_hip_debug.h, the hazard probes in include/nws_probe.h (a library of their own, ABI v6),
    the product ABI in include/nws_hip.h"""
    product, debug, probe = (_declared_symbols((h,)) for h in ("nws_hip.h", "nws_hip_debug.h", "nws_probe.h"))
    assert not [n for n in product if n.startswith(("nws_debug_", "nws_coexec_"))]
    assert debug and all(n.startswith("nws_debug_") for n in debug)
    assert probe and all(n.startswith("nws_coexec_") for n in probe)


def test_probe_library_is_apart_from_the_product_library():
    """VERDICT r5 hygiene (b): the product library holds no kernel on the build guard's allow-list - the hazard probe is
    libnws_probe.so, which exports exactly what include/nws_probe.h declares and nothing of the product ABI."""
    declared = _declared_symbols(("nws_probe.h",))
    probe = C.CDLL(_lib.PROBE_LIB_PATH)
    for name in declared:
        assert hasattr(probe, name), name
    assert sorted(_lib.PROBE_SYMBOLS) == declared
    product = C.CDLL(_lib.LIB_PATH)
    assert not any(hasattr(product, name) for name in declared)
    assert not hasattr(probe, "nws_forward")


def test_library_exports_every_declared_symbol():
    declared = _declared_symbols()
    assert len(declared) >= 25
    handle = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/nws_hip.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def test_abi_version_and_error_strings():
    L = _lib.lib()
    assert L.nws_abi_version() == _lib.ABI_VERSION == 6
    assert b"unsupported" in L.nws_error_string(-1)
    assert b"bad argument" in L.nws_error_string(-2)
    assert L.nws_error_string(0) == b"ok"


def test_struct_layouts_match_header_sizes():
    # the library reports sizeof() of its own structs: the ctypes declarations must agree byte for byte
    L = _lib.lib()
    assert C.sizeof(_lib.NwsReverbPlan) == 32 == L.nws_sizeof(1)
    assert C.sizeof(_lib.NwsWeights) == L.nws_sizeof(0)
    assert C.sizeof(_lib.NwsForwardAux) == L.nws_sizeof(2)
    n_ptr = sum(1 for _, t in _lib.NwsWeights._fields_ if t is _lib._fp) + 4 * 4 + 3 * 4
    assert C.sizeof(_lib.NwsWeights) == 8 * n_ptr + 16 + 8  # + lut_size, lut_min, lut_max (+4 pad), exciter_opts (+4 pad)


@pytest.mark.parametrize("N,L,N1,N2", [(64000, 64000, 125, 512), (256, 32000, 125, 256), (32000, 32000, 125, 256),
                                       (128 * 1000, 128000, 125, 1024), (128 * 2000, 256000, 125, 2048),
                                       (128 * 504, 64512, 63, 1024), (128 * 512, 65536, 32, 2048)])
def test_reverb_plan_is_host_only(N, L, N1, N2):
    plan = _lib.NwsReverbPlan()
    assert _lib.lib().nws_reverb_plan(N, 32000, C.byref(plan)) == 0
    assert (plan.L, plan.N1, plan.N2, plan.Lc, plan.hist, plan.nblk) == (L, N1, N2, 0, 0, 1)     # direct: one transform IS the circular convolution
    assert _lib.lib().nws_reverb_workspace_bytes(C.byref(plan), 3) == max(2 * 2 * L, 3 * L) * 4
    assert _lib.lib().nws_reverb_spectrum_bytes(C.byref(plan)) == 3 * L * 4   # Sre | Sim | [0, ir] in the time domain (short-buffer form)
    assert _lib.lib().nws_reverb_plan_serves(C.byref(plan), N, 32000) == 1


def test_every_clip_length_has_a_reverb_plan():
    """Reverb.forward takes any N (shaping.py:161-173); VERDICT r3 #1: lengths whose odd part exceeded 8192 were refused.  Every
    N = 128 T now plans - direct when the circular length is 125 x 2^k or (<= 128) x 2^k, else overlap-save on 125 x 2^k blocks
    that cover every output exactly once and keep a sensible share of each transform."""
    L = _lib.lib()
    plan = _lib.NwsReverbPlan()
    Ts = list(range(2, 4100)) + [8191, 8192, 8193, 9375, 10001, 16384, 20000, 37500, 37501, 65537, 131074, 1 << 20, (1 << 23)]
    for T in Ts:
        N = 128 * T
        assert L.nws_reverb_plan(N, 32000, C.byref(plan)) == 0, T
        assert plan.L == plan.N1 * plan.N2 and L.nws_reverb_plan_serves(C.byref(plan), N, 32000) == 1
        if plan.Lc == 0:
            assert plan.L == max(N, 32000) and plan.nblk == 1 and (plan.N1 == 125 or plan.N1 <= 128)
        else:
            P = plan.L - plan.hist
            assert plan.Lc == max(N, 32000) and plan.hist == 31999 and plan.N1 == 125
            assert (plan.nblk - 1) * P < N <= plan.nblk * P and 4 * P >= plan.L
            assert L.nws_reverb_workspace_bytes(C.byref(plan), 4) == 2 * 2 * plan.nblk * plan.L * 4
        assert L.nws_reverb_plan_serves(C.byref(plan), N + 128, 32000) == (1 if plan.Lc == 0 and N + 128 <= plan.L else 0)
    # other impulse-response lengths (runtime-size path): even circular lengths always plan, odd ones never (the reference's
    # rfft / irfft pair is not a circular convolution there: generators of csrc/generic.hip)
    assert L.nws_reverb_plan(100, 88200, C.byref(plan)) == 0 and (plan.L, plan.Lc, plan.hist, plan.nblk) == (128000, 88200, 88199, 1)
    assert L.nws_reverb_plan(441000, 44100, C.byref(plan)) == 0 and plan.Lc == 441000 and plan.hist == 44099
    assert L.nws_reverb_plan(1000, 1030, C.byref(plan)) == 0 and (plan.L, plan.Lc) == (32000, 1030)
    assert L.nws_reverb_plan(1001, 900, C.byref(plan)) == -1
    assert L.nws_reverb_plan(100, 1029, C.byref(plan)) == -1


def test_reverb_plan_rejects_bad_arguments():
    plan = _lib.NwsReverbPlan()
    assert _lib.lib().nws_reverb_plan(0, 32000, C.byref(plan)) == -2
    assert _lib.lib().nws_reverb_plan(64000, 0, C.byref(plan)) == -2
    assert _lib.lib().nws_reverb_plan((1 << 30) + 2, 32000, C.byref(plan)) == -2


def test_bad_arguments_return_codes_without_touching_the_gpu():
    L = _lib.lib()
    assert L.nws_phase_carry(None, None, 1, 4, None, None) == -2
    assert L.nws_control_gru(None, None, 1, 2, 4, None, None) == -2
    assert L.nws_fir_noise(None, None, None, 1, 4, None, None) == -2
    assert L.nws_forward_workspace_bytes(None, 1, 4) == 0


def test_gin_reader_parses_reference_and_packaged_config():
    gin = nws.gin
    for path in ([REF_GIN] if os.path.exists(REF_GIN) else []) + [nws.DEFAULT_GIN]:
        gin.clear_config()
        gin.parse_config_file(path)
        assert gin.query_parameter("Reverb.sr") == 16000
        assert gin.query_parameter("HarmonicOscillator.n_harmonics") == 101
        assert gin.query_parameter("NEWT.shaping_fn_size") == 8
        assert gin.query_parameter("noise_synth/TimeDistributedMLP.out_size") == 129
        assert gin.query_parameter("Reverb.length_in_seconds") == 2


def test_gin_scopes_macros_and_references():
    gin = nws.gin

    @gin.configurable
    class Thing:
        def __init__(self, a, b=1, c=None):
            self.a, self.b, self.c = a, b, c

    @gin.configurable("helper_fn")
    def helper(x=0):
        return x * 2

    gin.parse_config("""
        base = 7
        Thing.a = %base
        Thing.b = [1, 2.5, 'x']   # trailing comment
        inner/Thing.a = -3
        helper_fn.x = 21
        Thing.c = @helper_fn()
    """)
    t = Thing()
    assert (t.a, t.b, t.c) == (7, [1, 2.5, "x"], 42)
    with gin.config_scope("inner"):
        assert Thing().a == -3 and Thing().b == [1, 2.5, "x"]
    assert Thing(a=5).a == 5          # explicit arguments win
    with pytest.raises(TypeError):
        gin.parse_config("Thing.nope = 1")
        Thing()


def test_module_surface_and_state_dict_keys(weights):
    nws.gin.clear_config()
    nws.ensure_default_config()
    m = nws.NeuralWaveshaping()
    sd = m.state_dict()
    assert set(sd) == set(weights), set(sd) ^ set(weights)
    for k, v in weights.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        assert sd[k].dtype == torch.from_numpy(np.asarray(v)).dtype, k
    assert sum(p.numel() for p in m.parameters()) == 266945
    assert (m.sample_rate, m.control_hop) == (16000, 128)
    for attr in ("embedding", "osc", "harmonic_mixer", "newt", "h_generator", "noise_synth", "reverb"):
        assert hasattr(m, attr)
    with pytest.raises(AssertionError):
        nws.TimeDistributedMLP(8, 8, 8, depth=2)


def test_checkpoint_loading_npz_and_lightning_ckpt(weights):
    m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(ROOT, "tests", "golden", "weights_vn.npz"))
    for k, v in weights.items():
        assert np.array_equal(m.state_dict()[k].numpy(), v), k
    ck = "/root/reference/checkpoints/nws/vn/last.ckpt"
    if os.path.exists(ck):   # build container only; the GPU box has no reference tree
        m2 = nws.NeuralWaveshaping.load_from_checkpoint(ck)
        for k, v in weights.items():
            assert np.array_equal(m2.state_dict()[k].numpy(), v), k
        assert m2.hparams["n_waveshapers"] == 64


def test_no_cpu_fallback_anywhere():
    m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(ROOT, "tests", "golden", "weights_vn.npz"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 4), torch.zeros(1, 2, 4))
    with pytest.raises(RuntimeError):
        m.render_exciter(torch.zeros(1, 1, 512))
    with pytest.raises(RuntimeError):
        m.get_embedding(torch.zeros(1, 2, 4))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            nws.FastNEWT(m.newt)
    # the sub-modules run stand-alone stage kernels when called on their own (public L2 interface): CPU tensors
    # must raise, never compute
    calls = {m.embedding: (torch.zeros(1, 2, 4),), m.osc: (torch.zeros(1, 512),), m.h_generator: (torch.zeros(1, 128, 4),),
             m.newt: (torch.zeros(1, 64, 512), torch.zeros(1, 128, 4)), m.noise_synth: (torch.zeros(1, 129, 4),),
             m.reverb: (torch.zeros(1, 512),), m.newt.shaping_fn: (torch.zeros(1, 64, 8),), m.newt.waveshaping_index: (torch.zeros(2), ) * 3,
             m.newt.mlp.net[1]: (torch.zeros(1, 128, 4),)}
    for sub, args in calls.items():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            sub(*args)
    # ... and a copied / pickled model is a working model with its own engine (the pointer cache is not carried over)
    import copy
    import io
    m2 = copy.deepcopy(m)
    assert m2._engine is not m._engine and m2._engine._model_ref is m2
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert m3._engine._model_ref is m3 and torch.equal(m3.reverb.ir, m.reverb.ir)


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "neural-waveshaping-synthesis_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("oracle's", "").lower() or f in ("parallel.py",), (dirpath, f)


def test_normalisation_front_end(tmp_path):
    """checkpoint.load_normalisation / make_control (colab cells 6 and 15): control = (x - mean) / std with
    rows 0 (F0) and 1 (loudness) of data_mean / data_std, F0 itself handed to the model in Hz."""
    ck = importlib.import_module("neural-waveshaping-synthesis_amd.checkpoint")
    rng = np.random.default_rng(3)
    mean, std = rng.normal(size=(19, 1)).astype(np.float32), rng.uniform(0.5, 2, size=(19, 1))
    np.save(tmp_path / "data_mean.npy", mean)
    np.save(tmp_path / "data_std.npy", std)
    m, s = ck.load_normalisation(str(tmp_path))
    assert m.shape == (19,) and s.dtype == np.float64
    f0_hz, loud = 440.0 + rng.normal(size=50), rng.uniform(0, 1, size=50)
    f0, control = ck.make_control(f0_hz, loud, m, s)
    assert f0.shape == (1, 50) and control.shape == (2, 50) and f0.dtype == torch.float32
    assert np.allclose(f0.numpy()[0], f0_hz, rtol=1e-6)
    assert np.allclose(control.numpy()[0], (f0_hz - mean[0, 0]) / std[0, 0], rtol=1e-5)
    assert np.allclose(control.numpy()[1], (loud - mean[1, 0]) / std[1, 0], rtol=1e-5)
    ref = "/root/reference/checkpoints/nws/vn"
    if os.path.exists(ref):
        m2, s2 = ck.load_normalisation(ref)
        w = np.load(os.path.join(ROOT, "tests", "golden", "weights_vn.npz"))
        assert np.allclose(m2[:2], w["__data_mean__"]) and np.allclose(s2[:2], w["__data_std__"])


def test_loudness_front_end_binds_like_the_reference_and_has_no_cpu_fallback():
    """data/utils/loudness_extraction.py mirrors the reference module path, signature and gin binding names
    (gin/data/urmp_4second_crepe.gin:11-14); without a GPU it must raise, not compute on the host."""
    import inspect

    import numpy as np
    import pytest
    import torch

    nws = importlib.import_module("neural-waveshaping-synthesis_amd")
    le = importlib.import_module("neural-waveshaping-synthesis_amd.data.utils.loudness_extraction")
    sig = inspect.signature(le.extract_perceptual_loudness)
    assert list(sig.parameters) == ["audio", "sample_rate", "n_fft", "hop_length", "window", "epsilon", "interpolate_fn", "normalise"]
    assert sig.parameters["n_fft"].default == 2048 and sig.parameters["hop_length"].default == 512
    up = importlib.import_module("neural-waveshaping-synthesis_amd.data.utils.upsampling")
    assert sig.parameters["interpolate_fn"].default is up.linear_interpolation      # the reference's default (:49)
    # linear_interpolation: frames -> (frames - 1) hop + window samples, then the centre padding comes off
    fr = np.arange(5, dtype=np.float64)
    full = up.linear_interpolation(fr, 8, 4)
    assert full.shape == (24,) and full[0] == 0.0 and full[-1] == 4.0 and np.all(np.diff(full) > 0)
    cut = up.linear_interpolation(fr, 8, 4, original_length=10)
    assert cut.shape == (10,) and np.allclose(cut, full[4:14])
    nws.gin.parse_config("""
control_hop = 128
extract_perceptual_loudness.n_fft = 1024
extract_perceptual_loudness.hop_length = %control_hop
""")
    assert nws.gin.query_parameter("extract_perceptual_loudness.n_fft") == 1024
    if not torch.cuda.is_available():
        with pytest.raises((RuntimeError, AssertionError)):
            le.extract_perceptual_loudness(np.zeros(4000, dtype=np.float32))


def test_build_guard_finds_swizzled_packed_forms():
    """The ISA guard of build.py (co-execution hazard, DESIGN.md 5.3, LABBOOK.md '5.2'): clean on every product object, and it does see
    the forms when they are there (the probe kernel contains them on purpose)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("nws_build", os.path.join(ROOT, "neural-waveshaping-synthesis_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build(verbose=False)
    assert "coexec_probe.hip" not in b.SOURCES and b.PROBE_SOURCES == ["coexec_probe.hip"]
    for src in b.SOURCES:
        obj = os.path.join(b.OBJ, src.replace(".hip", ".o"))
        assert os.path.exists(obj), obj
        assert b.check_packed_swizzles(obj, ()) == [], src        # product objects: nothing is allowed
    assert b.check_packed_swizzles(os.path.join(b.OBJ, "coexec_probe.o")) == []       # the probe: only its own three kernels
    found = b.check_packed_swizzles(os.path.join(b.OBJ, "coexec_probe.o"), ())
    first = [i for k, i in found if "pk_probe_kernel" in k]
    assert len(first) == 4 and all("pk_probe" in k for k, _ in found), found      # the 4 hazardous fp32 forms of probe 1


def test_build_guard_finds_valu_writes_in_front_of_matrix_reads():
    """Second ISA guard of build.py (round 4): a v_mfma must not read a VGPR that a vector instruction wrote less than two wait
    states earlier.  hipcc keeps that distance for what it schedules but not for inline asm - the wave-resident frame-MLP kernel
    returned wrong first products of every layer until its fp16 split stopped being inline asm.  Clean on every product object;
    sees the pattern in a listing that has it (the sequence the first build of that kernel contained)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("nws_build", os.path.join(ROOT, "neural-waveshaping-synthesis_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build(verbose=False)
    for src in b.SOURCES:
        assert b.check_valu_mfma_hazard(os.path.join(b.OBJ, src.replace(".hip", ".o"))) == [], src
    listing = """
0000000000001000 <bad_kernel>:
\tv_fma_mixlo_f16 v127, v123, -1.0, v16 op_sel_hi:[1,0,0]    // 000000001000: D3A1007F
\tv_fma_mixhi_f16 v127, v123, -1.0, v17 op_sel:[1,0,0] op_sel_hi:[1,0,0]// 000000001008: D3A2087F
\ts_cmpk_gt_u32 s20, 0xff                                    // 000000001010: B51400FF
\tv_mfma_f32_32x32x16_f16 v[0:15], v[18:21], v[124:127], v[0:15]// 000000001018: D3D50000
0000000000002000 <good_kernel>:
\tv_fma_mixhi_f16 v127, v123, -1.0, v17 op_sel:[1,0,0] op_sel_hi:[1,0,0]// 000000002008: D3A2087F
\ts_nop 1                                                    // 000000002010: BF800001
\tv_mfma_f32_32x32x16_f16 v[0:15], v[18:21], v[124:127], v[0:15]// 000000002018: D3D50000
\tv_add_f32_e32 v40, v41, v42                                // 000000002020: 02505529
\tv_mfma_f32_32x32x16_f16 v[0:15], v[18:21], v[124:127], v[0:15]// 000000002028: D3D50000
"""
    b.device_disassembly = lambda obj: listing
    found = b.check_valu_mfma_hazard("unused")
    assert [k for k, _, _ in found] == ["bad_kernel"] and "v_fma_mixhi_f16 v127" in found[0][1], found


def test_build_guard_register_budgets():
    """Third guard of build.py (round 4): kernels whose register count decides how many workgroups a CU holds have a budget; a
    build that outgrows it fails instead of silently dropping a workgroup (the oscillator kernel ran two instead of three
    8-wave workgroups per CU at 93 registers).  Parses hipcc's resource remarks; sees an overrun in a listing that has one, and
    every budgeted kernel of the real build is found and inside its line."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("nws_build", os.path.join(ROOT, "neural-waveshaping-synthesis_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    remarks = """
x.hip:1:1: remark: Function Name: _ZN3foo19exciter_newt_kernelILi4ELi0ELi2ELi34EEEvPf [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     TotalSGPRs: 43 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     VGPRs: 93 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     AGPRs: 0 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark: Function Name: _ZN3foo26g_exciter_newt_mfma_kernelILi2ELi1EEEvPf [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     VGPRs: 96 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     AGPRs: 32 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark: Function Name: _ZN3foo12other_kernelEvPf [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     VGPRs: 250 [-Rpass-analysis=kernel-resource-usage]
"""
    assert b.kernel_registers(remarks) == {"_ZN3foo19exciter_newt_kernelILi4ELi0ELi2ELi34EEEvPf": 93,
                                           "_ZN3foo26g_exciter_newt_mfma_kernelILi2ELi1EEEvPf": 128, "_ZN3foo12other_kernelEvPf": 250}
    over = b.check_register_budgets(remarks)
    assert [("exciter_newt_kernelILi4ELi0ELi2ELi34E" in k, n, bud) for k, n, bud, _ in over] == [(True, 93, 80)], over
    # the real translation units: every budget line matches at least one kernel, and none is exceeded
    seen = set()
    for src in ("exciter_newt.hip", "generic.hip"):
        cmd = [b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), "-c", os.path.join(b.CSRC, src), "-o", os.devnull]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert b.check_register_budgets(r.stderr) == [], src
        seen |= {key for key, _, _ in b.REGISTER_BUDGETS for k in b.kernel_registers(r.stderr) if key in k}
    assert seen == {key for key, _, _ in b.REGISTER_BUDGETS}, seen


def test_precision_rule_is_a_worst_case_bound_from_the_weights():
    """Engine.exciter_opts' automatic choice (precision.hybrid_w_error_bound): host logic, no GPU.  The shipped checkpoint's bound
    is far above 1e-5 (-> every mixer product two-term); it scales linearly with ||ir||_1 and with the high-harmonic weights
    and vanishes when those weights are zero (-> the cheaper products are admitted)."""
    import importlib

    from oracle.newt_oracle import OracleNEWT, load_weights_npz

    prec = importlib.import_module("neural-waveshaping-synthesis_amd.precision")
    path = os.path.join(ROOT, "tests", "golden", "weights_vn.npz")
    w = {k: v for k, v in load_weights_npz(path).items() if not k.startswith("__")}
    table = OracleNEWT(w, fast=True).lookup_table()
    m = nws.NeuralWaveshaping.load_from_checkpoint(path)
    b0 = prec.hybrid_w_error_bound(m, table)
    assert b0["bound"] > 1e3 and b0["reverb_gain"] > 500
    with torch.no_grad():
        m.reverb.ir.mul_(0.5)
    b1 = prec.hybrid_w_error_bound(m, table)
    assert abs(b1["bound"] / b0["bound"] - (1 + 0.5 * (b0["reverb_gain"] - 1)) / b0["reverb_gain"]) < 1e-9
    with torch.no_grad():
        m.harmonic_mixer.weight[:, 15:].mul_(0.25)
    b2 = prec.hybrid_w_error_bound(m, table)
    assert abs(b2["bound"] / b1["bound"] - 0.25) < 1e-3
    with torch.no_grad():
        m.harmonic_mixer.weight[:, 15:].zero_()
    assert prec.hybrid_w_error_bound(m, table)["bound"] == 0.0
    g = prec.film_gain_bounds(m.newt.mlp)
    assert g.shape == (256,) and float(g.min()) > 0


def test_stream_bookkeeping_functions_are_consistent():
    """Host-side helpers of the streaming ABI (csrc/stream.hip; no GPU needed): whatever the chunking, the emitted samples
    add up to 128 F and the noise draws to the one-shot draw's 128 F - 1 samples plus the two look-ahead samples."""
    L = _lib.lib()
    for chunks in ([60], [1, 7, 16, 4, 31, 1], [2] * 30, [13, 47], [249, 51]):
        seen, emitted, drawn, prev_start = 0, 0, 0, 0
        for i, K in enumerate(chunks):
            first, final = int(seen == 0), int(i == len(chunks) - 1)
            emitted += L.nws_stream_out_samples(K, first, final)
            drawn += L.nws_stream_noise_draws(K, first, seen)
            start = L.nws_stream_noise_start(first, seen)
            assert start >= prev_start and start % 128 == 0
            prev_start = start
            seen += K
        F = sum(chunks)
        assert emitted == 128 * F and drawn == 128 * (F - 1) + 129
    pm = __import__("importlib").import_module("neural-waveshaping-synthesis_amd.pipeline")
    assert pm.ForwardPipeline.row_blocks(64, 4) == [(0, 16), (16, 16), (32, 16), (48, 16)]
    assert pm.ForwardPipeline.row_blocks(64, 3) is None and pm.ForwardPipeline.row_blocks(8, 4) is None
