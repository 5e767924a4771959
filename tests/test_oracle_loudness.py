"""The loudness oracle (oracle/loudness_oracle.py) restates librosa 0.8.0, which is absent here.  Its STFT half is pinned
against torch.stft (an independent implementation of the same documented semantics: center=True, reflect padding,
periodic hann window); its dB half against hand-computed cases."""
import numpy as np
import pytest
import torch

from oracle import loudness_oracle as lo


def test_stft_magnitude_matches_torch_stft():
    g = np.random.default_rng(0)
    for n, n_fft, hop in ((4000, 1024, 128), (2049, 256, 64), (16000, 2048, 512), (777, 64, 7)):
        x = g.standard_normal(n)
        ref = torch.stft(torch.from_numpy(x), n_fft, hop, window=torch.hann_window(n_fft, periodic=True, dtype=torch.float64),
                         center=True, pad_mode="reflect", return_complex=True).abs().numpy()
        got = lo.stft_magnitude(x, n_fft, hop)
        assert got.shape == (n_fft // 2 + 1, 1 + n // hop) == ref.shape
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


def test_db_reference_max_floor_and_top_db():
    mag = np.array([[1.0, 0.1], [1e-3, 1e-7], [0.0, 10.0]])
    db = lo.amplitude_to_db_refmax(mag, amin=1e-5, top_db=80.0)
    # ref = 10 -> 0 dB; 1 -> -20; 0.1 -> -40; 1e-3 -> -80; below the floor / more than 80 dB down -> clipped at -80
    assert np.allclose(db, [[-20.0, -40.0], [-80.0, -80.0], [-80.0, 0.0]], atol=1e-9)
    quiet = lo.amplitude_to_db_refmax(np.full((3, 2), 1e-7), amin=1e-5)     # all below amin: everything at the reference
    assert np.allclose(quiet, 0.0)


def test_extract_perceptual_loudness_shapes_and_interpolation():
    g = np.random.default_rng(1)
    x = 0.1 * g.standard_normal(8000)
    l = lo.extract_perceptual_loudness(x, n_fft=1024, hop_length=128)
    assert l.shape == (63,) and np.all(l <= 1.0) and np.all(l >= 0.0)
    up = lo.extract_perceptual_loudness(x, n_fft=1024, hop_length=128, interpolate_fn=lo.linear_interpolation)
    assert up.shape == (8000,)
    # a pure tone: every frame has the same spectrum away from the edges -> flat loudness there
    t = np.arange(16000) / 16000.0
    tone = lo.extract_perceptual_loudness(np.sin(2 * np.pi * 440 * t), n_fft=1024, hop_length=128)
    assert np.ptp(tone[8:-8]) < 2e-3


def test_window_is_the_one_librosa_asks_scipy_for():
    """librosa.stft(window="hann") obtains its window from scipy.signal.get_window("hann", n_fft, fftbins=True)
    (librosa 0.8.0 filters.get_window); scipy IS in the image, so this half of the chain is pinned on the real dependency."""
    import scipy.signal

    for n_fft in (64, 256, 1024, 2048):
        w = scipy.signal.get_window("hann", n_fft, fftbins=True)
        assert np.abs(lo.hann_periodic(n_fft) - w).max() <= 1e-15


def test_stft_magnitude_matches_scipy_stft():
    """A second independent implementation of the same framing (scipy.signal.stft with boundary="even" = numpy's reflect
    padding by n_fft/2, no zero padding of the tail); scipy scales by 1 / sum(window), librosa does not."""
    import scipy.signal

    g = np.random.default_rng(3)
    for n, n_fft, hop in ((4096, 1024, 128), (2048, 256, 64), (16384, 2048, 512)):
        x = g.standard_normal(n)
        w = scipy.signal.get_window("hann", n_fft, fftbins=True)
        _, _, Z = scipy.signal.stft(x, window=w, nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft, boundary="even", padded=False,
                                    return_onesided=True)
        ref = np.abs(Z) * w.sum()
        got = lo.stft_magnitude(x, n_fft, hop)
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


def test_whole_feature_matches_a_third_party_port_of_the_librosa_calls():
    """librosa itself is absent from the image (parity of row f4 stays UNPINNED), but `transformers.audio_utils` is here: an
    independent, widely used numpy port of exactly the two librosa calls the reference makes - `spectrogram(power=1.0, center=True,
    pad_mode="reflect")` for |librosa.stft| and `amplitude_to_db(reference, min_value, db_range)` for librosa.amplitude_to_db(ref=np.max,
    amin, top_db=80).  The whole feature (STFT magnitude -> dB against the clip's maximum -> 80 dB floor -> mean over bins -> (L + 80) / 80)
    built from those two functions must agree with the oracle on tones, noise, a click in silence and an all-zero clip.  A third
    implementation agreeing is evidence, not a pin: the header of oracle/loudness_oracle.py keeps saying so."""
    au = pytest.importorskip("transformers.audio_utils")
    rng = np.random.default_rng(5)
    sr = 16000
    t = np.arange(2 * sr) / sr
    clips = {
        "tone": 0.4 * np.sin(2 * np.pi * 440.0 * t * (1 + 0.002 * np.sin(2 * np.pi * 5.0 * t))),
        "noise": 0.1 * rng.standard_normal(2 * sr),
        "click": np.concatenate([np.zeros(9000), [0.9], np.zeros(7000)]),
        "zeros": np.zeros(5000),
        "quiet_then_loud": np.concatenate([1e-4 * rng.standard_normal(8000), 0.5 * rng.standard_normal(8000)]),
    }
    for name, y in clips.items():
        for n_fft, hop in ((1024, 128), (2048, 512), (256, 100)):
            got = lo.extract_perceptual_loudness(y, sr, n_fft, hop)
            win = au.window_function(n_fft, "hann", periodic=True)
            mag = au.spectrogram(y.astype(np.float64), win, frame_length=n_fft, hop_length=hop, power=1.0, center=True, pad_mode="reflect",
                                 dtype=np.float64)
            ref = float(mag.max())
            if ref > 0.0:
                db = au.amplitude_to_db(mag, reference=ref, min_value=1e-5, db_range=80.0)
            else:      # reference must be > 0 there; librosa clamps it to amin like the oracle: every bin sits at 0 dB
                db = np.zeros_like(mag)
            want = (db.mean(axis=0) + 80.0) / 80.0
            assert got.shape == want.shape == (1 + y.size // hop,), (name, n_fft, hop, got.shape, want.shape)
            assert np.max(np.abs(got - want)) <= 1e-7, (name, n_fft, hop, float(np.max(np.abs(got - want))))   # (1.0 = 80 dB; measured 3e-9)
