"""The loudness oracle (oracle/loudness_oracle.py) restates librosa 0.8.0, which is absent here.  Its STFT half is pinned
against torch.stft (an independent implementation of the same documented semantics: center=True, reflect padding,
periodic hann window); its dB half against hand-computed cases."""
import numpy as np
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
