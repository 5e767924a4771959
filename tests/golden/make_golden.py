#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference).  Nothing from the
reference is copied: the outputs are data (inputs, RNG draws, expected outputs,
trained weights as flat arrays).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures written:
  weights_vn.npz      state_dict of checkpoints/nws/vn/last.ckpt as flat arrays + data_mean/std rows 0-1
  g1_realistic.npz    B=1,T=500, 440 Hz + vibrato, loudness sweep (normalised with vn stats): y_newt, y_fast
  g2_rand.npz         B=1,T=500, torch.rand inputs exactly as scripts/time_forward_pass.py:27-40
  g3_stages.npz       B=2,T=3 (N=384): every stage intermediate, exact and LUT shapers
  g4_stream.npz       T=2 (N=256) and T=32 (N=4096) end to end (reverb L=32000 branch)
  g5_lut.npz          LUT rows/checksum + out-of-range probes of FastNEWT.shaping_fn
  g6_highf0.npz       B=1,T=500, F0 1-2 kHz (cumsum ~1e8, large sine arguments)
  weights_fl.npz / weights_tpt.npz   the other two shipped instruments (checkpoints/nws/{fl,tpt}/last.ckpt), same layout
  g7_fl.npz / g7_tpt.npz             B=1,T=125 (1 s) realistic vector per instrument: y_newt, y_fast, draws
  g8_small / g8_odd / g8_oddlen .npz three NON-default gin configurations (random init, recorded weights + gin text): inputs, draws,
                                     stage taps, y_newt, y_fast   (`... make_golden.py generic` regenerates only these)
  g9_upsampling.npz   data/utils/upsampling.py: linear / cubic-spline / overlap-add interpolators (`... make_golden.py upsampling`)
  g10_timing_script.npz  the literal model of scripts/time_forward_pass.py:27-43: UNMODIFIED random-init NeuralWaveshaping() under
                         gin/models/newt.gin, torch.rand inputs at T = 500, exact and FastNEWT (`... make_golden.py timing`)
(`python tests/golden/make_golden.py instruments` regenerates only the fl / tpt four.)
The RNG draws made inside forward are recorded by wrapping torch.rand / torch.rand_like.
"""
import hashlib
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_refstubs"))
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import gin  # noqa: E402  (stub -> product ginlite)

gin.parse_config_file(os.path.join(REF, "gin/models/newt.gin"))
from neural_waveshaping_synthesis.models.neural_waveshaping import NeuralWaveshaping  # noqa: E402
from neural_waveshaping_synthesis.models.modules.shaping import FastNEWT  # noqa: E402

torch.set_grad_enabled(False)


class DrawRecorder:
    """Record the tensors returned by torch.rand / torch.rand_like during a forward."""

    def __enter__(self):
        self.draws = []
        self._rand, self._rand_like = torch.rand, torch.rand_like

        def rand(*a, **k):
            t = self._rand(*a, **k)
            self.draws.append(t.clone())
            return t

        def rand_like(*a, **k):
            t = self._rand_like(*a, **k)
            self.draws.append(t.clone())
            return t

        torch.rand, torch.rand_like = rand, rand_like
        return self

    def __exit__(self, *exc):
        torch.rand, torch.rand_like = self._rand, self._rand_like


def run(model, f0, control, seed):
    torch.manual_seed(seed)
    with DrawRecorder() as rec:
        y = model(f0, control)
    assert len(rec.draws) == 2 and rec.draws[0].shape == (1, 101, 1), [d.shape for d in rec.draws]
    return y, rec.draws[0].reshape(-1).numpy(), rec.draws[1].numpy()


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB  [{', '.join(out)}]")


def main():
    ck = os.path.join(REF, "checkpoints/nws/vn")
    model = NeuralWaveshaping.load_from_checkpoint(os.path.join(ck, "last.ckpt")).eval()
    exact_newt = model.newt
    fast_newt = FastNEWT(exact_newt)
    mean = np.load(os.path.join(ck, "data_mean.npy")).astype(np.float64)
    std = np.load(os.path.join(ck, "data_std.npy")).astype(np.float64)

    sd = {k: v.numpy() for k, v in model.state_dict().items() if not k.startswith("newt.shaping_fn_dummy")}
    sd = {k: v for k, v in model.state_dict().items()}
    save("weights_vn.npz", **{k: v for k, v in sd.items()},
         **{"__data_mean__": mean[:2, 0].astype(np.float32), "__data_std__": std[:2, 0].astype(np.float32)})

    def both(f0, control, seed):
        model.newt = exact_newt
        y_newt, pu, nz = run(model, f0, control, seed)
        model.newt = fast_newt
        y_fast, pu2, nz2 = run(model, f0, control, seed)
        assert np.array_equal(pu, pu2) and np.array_equal(nz, nz2)
        return y_newt, y_fast, pu, nz

    T = 500
    t = np.arange(T) * 128 / 16000.0
    # G1: realistic violin-range input, normalised like colab cell 15
    f0_hz = 440.0 * (1 + 0.01 * np.sin(2 * np.pi * 5.5 * t))
    loud = 0.35 + 0.3 * np.sin(np.pi * t / 4.0) ** 2
    f0 = torch.tensor(f0_hz, dtype=torch.float32).view(1, 1, T)
    control = torch.tensor(np.stack([(f0_hz - mean[0, 0]) / std[0, 0], (loud - mean[1, 0]) / std[1, 0]]),
                           dtype=torch.float32).view(1, 2, T)
    y_newt, y_fast, pu, nz = both(f0, control, 1234)
    save("g1_realistic.npz", f0=f0, control=control, phase_u=pu, noise=nz, y_newt=y_newt, y_fast=y_fast)
    shared_pu, shared_nz = pu, nz

    # G2: the reference timing script's inputs (values in [0,1): sub-1 Hz "F0")
    torch.manual_seed(7)
    control2 = torch.rand(1, 2, T)
    f02 = torch.rand(1, 1, T)
    y_newt, y_fast, pu, nz = both(f02, control2, 1234)
    assert np.array_equal(nz, shared_nz) and np.array_equal(pu, shared_pu)
    save("g2_rand.npz", f0=f02, control=control2, y_newt=y_newt, y_fast=y_fast)  # draws: same as g1 (same seed)

    # G6: high F0 (1-2 kHz glide): cumsum magnitude ~1e8, sine arguments ~1e6
    f0_hz6 = np.linspace(1000.0, 2000.0, T) * (1 + 0.005 * np.sin(2 * np.pi * 6.0 * t))
    f06 = torch.tensor(f0_hz6, dtype=torch.float32).view(1, 1, T)
    control6 = torch.tensor(np.stack([(f0_hz6 - mean[0, 0]) / std[0, 0], (loud - mean[1, 0]) / std[1, 0]]),
                            dtype=torch.float32).view(1, 2, T)
    y_newt, y_fast, pu, nz = both(f06, control6, 1234)
    save("g6_highf0.npz", f0=f06, control=control6, y_newt=y_newt, y_fast=y_fast)

    # G3: stage vectors, B=2, T=3
    Ts = 3
    torch.manual_seed(11)
    f0s = torch.stack([torch.tensor([[220.0, 233.1, 246.9]]), torch.tensor([[3950.0, 4100.0, 90.0]])])  # (2,1,3)
    controls = torch.randn(2, 5, Ts)  # extra channels must be ignored
    taps = {}

    def tap(name, what="out"):
        def hook(mod, inp, out):
            taps[name] = (inp[0] if what == "in" else out[0] if isinstance(out, tuple) else out).clone()
        return hook

    hs = [
        model.osc.register_forward_hook(tap("f0_up", "in")),
        model.osc.register_forward_hook(tap("osc")),
        model.harmonic_mixer.register_forward_hook(tap("exciter")),
        model.embedding.gru.register_forward_hook(tap("gru_out")),
        model.embedding.register_forward_hook(tap("embedding")),
        exact_newt.mlp.register_forward_hook(tap("film")),
        exact_newt.waveshaping_index.register_forward_hook(tap("lut_arg")),
        exact_newt.normalising_coeff.register_forward_hook(tap("shaped", "in")),
        model.h_generator.register_forward_hook(tap("H")),
        model.noise_synth.register_forward_hook(tap("noise_out")),
        model.reverb.register_forward_hook(tap("pre_reverb", "in")),
    ]
    model.newt = exact_newt
    h_newt = exact_newt.register_forward_hook(tap("newt_out"))
    y_exact, pu3, nz3 = run(model, f0s, controls, 99)
    exact_taps = dict(taps)
    h_newt.remove()
    model.newt = fast_newt
    h_newt = fast_newt.register_forward_hook(tap("newt_out"))
    y_lut, _, _ = run(model, f0s, controls, 99)
    for h in hs + [h_newt]:
        h.remove()
    save("g3_stages.npz", f0=f0s, control=controls, phase_u=pu3, noise=nz3,
         f0_up=exact_taps["f0_up"], osc=exact_taps["osc"], exciter=exact_taps["exciter"],
         gru_out=exact_taps["gru_out"], embedding=exact_taps["embedding"], film=exact_taps["film"],
         lut_arg=exact_taps["lut_arg"], shaped_exact=exact_taps["shaped"], shaped_lut=taps["shaped"],
         newt_out_exact=exact_taps["newt_out"][:, 0], newt_out_lut=taps["newt_out"][:, 0], H=exact_taps["H"],
         noise_out=exact_taps["noise_out"][:, 0], pre_reverb_exact=exact_taps["pre_reverb"],
         pre_reverb_lut=taps["pre_reverb"], y_exact=y_exact, y_lut=y_lut)

    # G4: streaming buffer sizes (scripts/time_buffer_sizes.py): N <= 32000 -> reverb L = 32000
    out = {}
    for Tn in (2, 32):
        torch.manual_seed(21 + Tn)
        f0n = 200.0 + 300.0 * torch.rand(2, 1, Tn)
        cn = torch.randn(2, 2, Tn)
        y_newt, y_fast, pu, nz = both(f0n, cn, 77)
        out.update({f"f0_T{Tn}": f0n, f"control_T{Tn}": cn, f"phase_u_T{Tn}": pu, f"noise_T{Tn}": nz,
                    f"y_newt_T{Tn}": y_newt, f"y_fast_T{Tn}": y_fast})
    save("g4_stream.npz", **out)

    # G5: the LUT itself and its out-of-range behaviour
    table = fast_newt.lookup_table.detach()
    probes = torch.tensor([-5.0, -3.0005, -3.0, -1.234567, 0.0, 0.3333333, 2.9985, 2.9993, 3.0, 4.0])
    xp = probes.view(1, 1, -1).expand(1, 64, -1).contiguous()
    save("g5_lut.npz", rows=np.array([0, 17, 63]), table_rows=table[[0, 17, 63]],
         table_sum=np.float64(table.double().sum().item()),
         table_abs_sum=np.float64(table.double().abs().sum().item()),
         table_sha256=np.frombuffer(hashlib.sha256(table.numpy().tobytes()).digest(), dtype=np.uint8),
         probes=probes, probe_out=fast_newt.shaping_fn(xp)[0], probe_exact=exact_newt.shaping_fn(xp)[0])


def instruments():
    """fl and tpt: weights + one realistic 1 s vector each (usable on all three shipped instruments)."""
    T = 125
    t = np.arange(T) * 128 / 16000.0
    for inst, base_hz in (("fl", 587.33), ("tpt", 349.23)):
        ck = os.path.join(REF, "checkpoints/nws", inst)
        model = NeuralWaveshaping.load_from_checkpoint(os.path.join(ck, "last.ckpt")).eval()
        exact_newt = model.newt
        fast_newt = FastNEWT(exact_newt)
        mean = np.load(os.path.join(ck, "data_mean.npy")).astype(np.float64)
        std = np.load(os.path.join(ck, "data_std.npy")).astype(np.float64)
        sd = {k: v for k, v in model.state_dict().items()}
        save(f"weights_{inst}.npz", **sd,
             **{"__data_mean__": mean[:2, 0].astype(np.float32), "__data_std__": std[:2, 0].astype(np.float32)})
        f0_hz = base_hz * (1 + 0.012 * np.sin(2 * np.pi * 5.0 * t)) * np.where(t < 0.5, 1.0, 2 ** (2 / 12))
        loud = 0.3 + 0.35 * np.sin(np.pi * t / 1.0) ** 2
        f0 = torch.tensor(f0_hz, dtype=torch.float32).view(1, 1, T)
        control = torch.tensor(np.stack([(f0_hz - mean[0, 0]) / std[0, 0], (loud - mean[1, 0]) / std[1, 0]]),
                               dtype=torch.float32).view(1, 2, T)
        model.newt = exact_newt
        y_newt, pu, nz = run(model, f0, control, 4321)
        model.newt = fast_newt
        y_fast, pu2, nz2 = run(model, f0, control, 4321)
        assert np.array_equal(pu, pu2) and np.array_equal(nz, nz2)
        save(f"g7_{inst}.npz", f0=f0, control=control, phase_u=pu, noise=nz, y_newt=y_newt, y_fast=y_fast)


GENERIC_GINS = {
    # every size the reference's gin surface names, away from gin/models/newt.gin (incl. the code defaults shaping_fn_size 16,
    # TrainableNonlinearity.depth 3 of shaping.py:17,46), two NEWT output channels, hop 64, 128-tap noise FIR, 1 s reverb
    "g8_small": """
Reverb.sr = 16000
Reverb.length_in_seconds = 1
noise_synth/FIRNoiseSynth.hop_length = 64
noise_synth/FIRNoiseSynth.ir_length = 128
noise_synth/TimeDistributedMLP.depth = 3
noise_synth/TimeDistributedMLP.out_size = 65
noise_synth/TimeDistributedMLP.hidden_size = 64
noise_synth/TimeDistributedMLP.in_size = 80
NEWT.out_channels = 2
NEWT.control_embedding_size = 80
NEWT.n_waveshapers = 32
HarmonicOscillator.sample_rate = 16000
HarmonicOscillator.n_harmonics = 60
ControlModule.embedding_size = 80
ControlModule.hidden_size = 96
ControlModule.control_size = 2
NeuralWaveshaping.sample_rate = 16000
NeuralWaveshaping.control_hop = 64
NeuralWaveshaping.n_waveshapers = 32
""",
    # odd sizes nothing divides: 7 harmonics, 5 shapers of width 3 / depth 2, GRU 33, embedding 17, hop 10, 30-tap FIR, and a
    # reverb whose circular length (1000) has no power-of-two factor >= 32 (time-domain reverb), 8 kHz sample rate
    "g8_odd": """
Reverb.sr = 1000
Reverb.length_in_seconds = 1
noise_synth/FIRNoiseSynth.hop_length = 10
noise_synth/FIRNoiseSynth.ir_length = 30
noise_synth/TimeDistributedMLP.depth = 4
noise_synth/TimeDistributedMLP.out_size = 16
noise_synth/TimeDistributedMLP.hidden_size = 20
noise_synth/TimeDistributedMLP.in_size = 17
TrainableNonlinearity.depth = 2
NEWT.shaping_fn_size = 3
NEWT.out_channels = 1
NEWT.control_embedding_size = 17
NEWT.n_waveshapers = 5
HarmonicOscillator.sample_rate = 8000
HarmonicOscillator.n_harmonics = 7
ControlModule.embedding_size = 17
ControlModule.hidden_size = 33
ControlModule.control_size = 2
NeuralWaveshaping.sample_rate = 8000
NeuralWaveshaping.control_hop = 10
NeuralWaveshaping.n_waveshapers = 5
""",
    # found by the random-configuration sweep of tests/test_gpu_generic.py: a reverb of an ODD number of samples (rfft of 1029
    # points, irfft back to 1028: shaping.py:171-173 passes no length) and a hop that is not a power of two with F0 up to 0.3 sr
    # over 40 frames (the upsampling's fused index arithmetic decides single ulps of a phase that reaches thousands of radians)
    "g8_oddlen": """
Reverb.sr = 1029
Reverb.length_in_seconds = 1
noise_synth/FIRNoiseSynth.hop_length = 25
noise_synth/FIRNoiseSynth.ir_length = 60
noise_synth/TimeDistributedMLP.depth = 3
noise_synth/TimeDistributedMLP.out_size = 31
noise_synth/TimeDistributedMLP.hidden_size = 12
noise_synth/TimeDistributedMLP.in_size = 21
TrainableNonlinearity.depth = 3
NEWT.shaping_fn_size = 5
NEWT.out_channels = 2
NEWT.control_embedding_size = 21
NEWT.n_waveshapers = 9
HarmonicOscillator.sample_rate = 8000
HarmonicOscillator.n_harmonics = 24
ControlModule.embedding_size = 21
ControlModule.hidden_size = 25
ControlModule.control_size = 2
NeuralWaveshaping.sample_rate = 8000
NeuralWaveshaping.control_hop = 25
NeuralWaveshaping.n_waveshapers = 9
""",
}


def generic_configs():
    """Non-default gin configurations: random-init reference models built from the gin text above, their
    state dicts, inputs, the two recorded draws, stage taps and outputs for exact NEWT and FastNEWT."""
    import json

    for name, text in GENERIC_GINS.items():
        gin.clear_config()
        gin.parse_config(text)
        torch.manual_seed(2024)
        model = NeuralWaveshaping().eval()
        K, hop = model.osc.n_harmonics, model.control_hop
        with torch.no_grad():
            # random initialisation leaves reverb.ir at 1e-6 and the LUT argument partly outside [-3, 3] (where the reference
            # extrapolates with |fract| ~ 1e3): give the IR audible energy and keep the index FiLM inside the table
            model.reverb.ir.copy_(torch.randn_like(model.reverb.ir) * 0.05 * torch.exp(-torch.arange(model.reverb.ir.numel()) / 300.0))
            S = model.newt.n_waveshapers
            model.newt.mlp.net[9].weight[:2 * S] *= 0.5
            model.newt.mlp.net[9].bias[:2 * S] *= 0.5
            model.newt.shaping_fn.input_scale.mul_(0.3)
        exact_newt = model.newt
        fast_newt = FastNEWT(exact_newt, table_size=4096 if name == "g8_small" else 1024, table_min=-3.0 if name == "g8_small" else -4.0)
        B, T = {"g8_small": (2, 16), "g8_odd": (3, 50), "g8_oddlen": (2, 40)}[name]
        torch.manual_seed(5)
        f0 = 150.0 + 500.0 * torch.rand(B, 1, 1) + 30.0 * torch.randn(B, 1, T).cumsum(-1) / 4
        if name == "g8_oddlen":
            f0 = 100.0 + 0.3 * model.sample_rate * torch.rand(B, 1, 1) * torch.rand(B, 1, T)
        else:
            f0[-1, 0, T // 2:] = 0.6 * model.sample_rate                   # above Nyquist: every harmonic masked
        control = torch.randn(B, 3, T)                                     # one extra channel, ignored (:70-71)
        taps = {}

        def tap(nm, what="out"):
            def hook(mod, inp, out):
                taps[nm] = (inp[0] if what == "in" else out[0] if isinstance(out, tuple) else out).clone()
            return hook

        hs = [model.osc.register_forward_hook(tap("osc")), model.harmonic_mixer.register_forward_hook(tap("exciter")),
              model.embedding.register_forward_hook(tap("embedding")), exact_newt.mlp.register_forward_hook(tap("film")),
              model.h_generator.register_forward_hook(tap("H")), model.noise_synth.register_forward_hook(tap("noise_out")),
              model.reverb.register_forward_hook(tap("pre_reverb", "in")), exact_newt.register_forward_hook(tap("newt_out"))]

        def run_g(seed):
            torch.manual_seed(seed)
            with DrawRecorder() as rec:
                y = model(f0, control)
            assert len(rec.draws) == 2 and rec.draws[0].shape == (1, K, 1) and rec.draws[1].shape == (hop * T - 1,)
            return y, rec.draws[0].reshape(-1).numpy(), rec.draws[1].numpy()

        model.newt = exact_newt
        y_newt, pu, nz = run_g(31)
        et = dict(taps)
        for h in hs:
            h.remove()
        model.newt = fast_newt
        y_fast, pu2, nz2 = run_g(31)
        assert np.array_equal(pu, pu2) and np.array_equal(nz, nz2)
        model.newt = exact_newt
        sd = {k: v for k, v in model.state_dict().items()}
        hp = dict(n_waveshapers=int(model.newt.n_waveshapers), control_hop=int(hop), sample_rate=float(model.sample_rate))
        save(f"{name}.npz", **sd, __gin__=np.array(text), __hparams__=np.array(json.dumps(hp)),
             __table_size__=np.int64(fast_newt.table_size), __table_min__=np.float64(fast_newt.table_min),
             __table_max__=np.float64(fast_newt.table_max), __lookup_table__=fast_newt.lookup_table.detach(),
             __f0__=f0, __control__=control, __phase_u__=pu, __noise__=nz, __y_newt__=y_newt, __y_fast__=y_fast,
             __osc__=et["osc"], __exciter__=et["exciter"], __embedding__=et["embedding"], __film__=et["film"], __H__=et["H"],
             __noise_out__=et["noise_out"], __pre_reverb__=et["pre_reverb"], __newt_out__=et["newt_out"])
    gin.clear_config()


def upsampling_vectors():
    """g9_upsampling.npz: the three frame-rate -> sample-rate interpolators of data/utils/upsampling.py (the gin-selectable
    `interpolate_fn` of the loudness / F0 features) on seeded inputs, with and without `original_length`."""
    from neural_waveshaping_synthesis.data.utils import upsampling as ref_up
    rng = np.random.default_rng(9)
    out = {}
    for i, (frames, win, hop, orig) in enumerate([(12, 256, 64, 0), (30, 512, 128, 3500), (7, 256, 64, 300), (33, 400, 100, 3111)]):
        sig = rng.standard_normal(frames).cumsum()
        out[f"c{i}_args"] = np.array([frames, win, hop, orig], dtype=np.int64)
        out[f"c{i}_signal"] = sig
        kw = dict(original_length=orig) if orig else {}
        out[f"c{i}_linear"] = ref_up.linear_interpolation(sig, win, hop, **kw)
        out[f"c{i}_cubic"] = ref_up.cubic_spline_interpolation(sig, win, hop, **kw)
        out[f"c{i}_ola"] = ref_up.overlap_add_upsample(sig, win, hop, **kw)
        out[f"c{i}_ola_tri3"] = ref_up.overlap_add_upsample(sig, win, hop, window_fn="triang", window_scale=3, **kw)
    np.savez(os.path.join(HERE, "g9_upsampling.npz"), **out)
    print("g9_upsampling.npz", {k: v.shape for k, v in out.items() if k.startswith("c1_")})


def timing_script_model():
    """g10_timing_script.npz: what scripts/time_forward_pass.py really runs (BASELINE config 1) - `NeuralWaveshaping()` with no
    checkpoint (:41), `model.newt = FastNEWT(model.newt)` (:43), `torch.rand` control / f0 (:27-40) - with fixed seeds: the random
    initial weights, the inputs, the two recorded draws and the reference's outputs for both shapers."""
    gin.clear_config()
    gin.parse_config_file(os.path.join(REF, "gin/models/newt.gin"))
    torch.manual_seed(41)
    control = torch.rand(1, 2, 16000 * 4 // 128)                          # :27-33
    f0 = torch.rand(1, 1, 16000 * 4 // 128)                               # :34-40
    model = NeuralWaveshaping()                                           # :41
    exact_newt = model.newt
    model.eval()
    y_newt, pu, nz = run(model, f0, control, 4141)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.newt = FastNEWT(exact_newt)                                     # :43
    y_fast, pu2, nz2 = run(model, f0, control, 4141)
    assert np.array_equal(pu, pu2) and np.array_equal(nz, nz2)
    save("g10_timing_script.npz", **sd, __f0__=f0, __control__=control, __phase_u__=pu, __noise__=nz, __y_newt__=y_newt, __y_fast__=y_fast)


if __name__ == "__main__":
    if sys.argv[1:] == ["timing"]:
        timing_script_model()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "upsampling":
        upsampling_vectors()
        sys.exit(0)
    if sys.argv[1:] == ["instruments"]:
        instruments()
    elif sys.argv[1:] == ["generic"]:
        generic_configs()
    else:
        main()
        instruments()
        generic_configs()
        timing_script_model()
