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
(`python tests/golden/make_golden.py instruments` regenerates only the last four.)
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


if __name__ == "__main__":
    if sys.argv[1:] == ["instruments"]:
        instruments()
    else:
        main()
        instruments()
