"""-m gpu: the HIP loudness feature (csrc/loudness.hip through the C-ABI) against the oracle (float64 restatement of
librosa 0.8.0's stft / amplitude_to_db; see oracle/loudness_oracle.py for what pins it)."""
import numpy as np
import pytest
import torch

from gpu_util import record

pytestmark = pytest.mark.gpu


def _cases():
    g = np.random.default_rng(5)
    t = np.arange(64000) / 16000.0
    vib = 440.0 * (1 + 0.01 * np.sin(2 * np.pi * 5.5 * t))
    tone = 0.3 * np.sin(2 * np.pi * np.cumsum(vib) / 16000.0) * np.linspace(0.05, 1.0, t.size)
    return {
        "violin_like": (tone + 1e-3 * g.standard_normal(t.size), 1024, 128),
        "noise_short_odd_length": (0.2 * g.standard_normal(4099), 1024, 128),
        "default_args_2048_512": (0.2 * g.standard_normal(30000), 2048, 512),
        "silence_then_click": (np.concatenate([np.zeros(3000), [1.0], np.zeros(3000)]), 1024, 128),
        "all_zero": (np.zeros(5000), 1024, 128),
        "small_fft_hop_not_pow2": (0.5 * g.standard_normal(3001), 256, 100),
    }


@pytest.mark.parametrize("name", list(_cases()))
def test_loudness_matches_oracle(name):
    import nws_amd
    from nws_amd.data.utils.loudness_extraction import extract_perceptual_loudness
    from oracle import loudness_oracle as lo
    audio, n_fft, hop = _cases()[name]
    ref = lo.extract_perceptual_loudness(audio, n_fft=n_fft, hop_length=hop)
    got = extract_perceptual_loudness(audio.astype(np.float32), n_fft=n_fft, hop_length=hop, interpolate_fn=None)
    assert isinstance(got, np.ndarray) and got.shape == ref.shape == (1 + audio.size // hop,)
    err = float(np.abs(got - ref).max())
    record("loudness_" + name, max_abs_err_normalised=err, frames=int(ref.size))
    # normalised units (1.0 = 80 dB).  fp32 transform against a float64 oracle: bins close to the -80 dB clip carry the
    # rounding noise of a 1024-term fp32 sum, the reference's own float32 FFT is in the same class
    assert err <= 2e-5, (name, err)


def test_loudness_batched_tensor_api_interpolation_and_errors():
    import nws_amd
    from nws_amd.data.utils.loudness_extraction import extract_perceptual_loudness, loudness_frames
    from oracle import loudness_oracle as lo
    g = np.random.default_rng(9)
    x = (0.1 * g.standard_normal((3, 8000))).astype(np.float32)
    x[1] *= 1e-3                                   # per-utterance reference maximum
    out = loudness_frames(torch.from_numpy(x).cuda(), 1024, 128)
    assert out.shape == (3, 63)
    for i in range(3):
        ref = lo.extract_perceptual_loudness(x[i].astype(np.float64), n_fft=1024, hop_length=128)
        assert np.abs(out[i].cpu().numpy() - ref).max() <= 2e-5
    raw = loudness_frames(torch.from_numpy(x).cuda(), 1024, 128, normalise=False).cpu().numpy()
    assert np.allclose((raw + 80) / 80, out.cpu().numpy(), atol=1e-6)
    up = extract_perceptual_loudness(x[0], n_fft=1024, hop_length=128, interpolate_fn=lo.linear_interpolation)
    ref_up = lo.extract_perceptual_loudness(x[0].astype(np.float64), n_fft=1024, hop_length=128, interpolate_fn=lo.linear_interpolation)
    assert up.shape == (8000,) and np.abs(up - ref_up).max() <= 2e-5
    # the reference's DEFAULT is sample-rate loudness (interpolate_fn=linear_interpolation, loudness_extraction.py:49)
    dflt = extract_perceptual_loudness(x[0], n_fft=1024, hop_length=128)
    assert dflt.shape == (8000,) and np.abs(dflt - ref_up).max() <= 2e-5
    # hop / n_fft pairs whose 31 hop + n_fft sample tile exceeds the 160 KB of LDS are refused with a clear message
    with pytest.raises(RuntimeError, match="LDS"):
        loudness_frames(torch.rand(1, 70000, device="cuda"), 2048, 2048)
    # throughput on the synthesis bench's shape (64 clips of 4 s) and on one 5-minute file
    big = torch.rand(64, 64000, device="cuda") - 0.5
    long1 = torch.rand(1, 16000 * 300, device="cuda") - 0.5
    for tag, a in (("64x4s", big), ("1x300s", long1)):
        loudness_frames(a, 1024, 128)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            loudness_frames(a, 1024, 128)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 5
        record("loudness_time_" + tag, ms=ms, x_realtime=a.numel() / 16000.0 / (ms * 1e-3))
    with pytest.raises(RuntimeError):
        loudness_frames(torch.from_numpy(x), 1024, 128)                     # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        loudness_frames(torch.from_numpy(x).cuda(), 1000, 128)              # n_fft not a power of two
    with pytest.raises(RuntimeError):
        loudness_frames(torch.from_numpy(x[:, :400]).cuda(), 1024, 128)     # shorter than the reflect padding
