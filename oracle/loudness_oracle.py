"""CPU restatement of the reference's perceptual-loudness feature.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else).  The product path is the HIP kernel behind
`neural-waveshaping-synthesis_amd/data/utils/loudness_extraction.py`.

Reference: neural_waveshaping_synthesis/data/utils/loudness_extraction.py
  compute_power_spectrogram   :10-22   librosa.stft -> np.abs -> librosa.amplitude_to_db(ref=np.max, amin=epsilon)
  perform_perceptual_weighting:25-38   A-weighting is computed but NOT applied (":38  weighted_spectrogram = power_spectrogram_in_db  # + weights")
  extract_perceptual_loudness :41-67   mean over bins, optional interpolate_fn, (L + 80) / 80
with gin/data/urmp_4second_crepe.gin:11-14 (n_fft 1024, hop 128, interpolate_fn None).

The arithmetic lives in librosa, a third-party dependency that is NOT in this image (requirements.txt:5 pins
librosa==0.8.0).  Its published algorithm, restated here:
  librosa.stft(y, n_fft, hop_length, window="hann")      core/spectrum.py (0.8.0): win_length = n_fft, window =
      scipy.signal.get_window("hann", n_fft, fftbins=True) (periodic), center=True with
      np.pad(y, n_fft // 2, mode="reflect"), frames = 1 + len(y) // hop_length, rfft of window * frame.
  librosa.amplitude_to_db(S, ref=np.max, amin, top_db=80) core/spectrum.py: magnitude = |S|; ref_value = ref(magnitude);
      power_to_db(magnitude**2, ref=ref_value**2, amin=amin**2, top_db): log_spec = 10 log10(max(amin^2, S^2))
      - 10 log10(max(amin^2, ref^2)); log_spec = max(log_spec, log_spec.max() - top_db).
Parity status: UNPINNED against librosa itself (absent, no network).  The STFT half is pinned against an independent
implementation with the same documented semantics (torch.stft, center=True, reflect, periodic hann) in
tests/test_oracle_loudness.py, against scipy.signal.stft (a second one), and the window against scipy.signal.get_window - the
very call librosa makes, scipy being in the image; the dB half is a direct transcription of the formulas above.  Round 5: the WHOLE
feature also agrees to 3e-9 (normalised units) with `transformers.audio_utils.spectrogram` + `amplitude_to_db` - a third-party numpy port
of the two librosa calls that IS in the image (tests/test_oracle_loudness.py) - on tones, noise, a click in silence and silence.  Evidence,
not a pin: librosa's own outputs have never been seen here.
"""
import numpy as np


def hann_periodic(n_fft: int) -> np.ndarray:
    n = np.arange(n_fft, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)


def stft_magnitude(audio: np.ndarray, n_fft: int, hop_length: int) -> np.ndarray:
    """|librosa.stft(audio, n_fft, hop_length, window='hann')|  ->  (1 + n_fft/2, frames), float64 arithmetic."""
    y = np.asarray(audio, dtype=np.float64)
    if y.ndim != 1:
        raise ValueError("audio must be 1-D")
    if y.size <= n_fft // 2:
        raise ValueError("reflect padding needs more than n_fft/2 samples")
    yp = np.pad(y, n_fft // 2, mode="reflect")
    frames = 1 + y.size // hop_length
    idx = hop_length * np.arange(frames)[None, :] + np.arange(n_fft)[:, None]
    spec = np.fft.rfft(hann_periodic(n_fft)[:, None] * yp[idx], axis=0)
    return np.abs(spec)


def amplitude_to_db_refmax(mag: np.ndarray, amin: float, top_db: float = 80.0) -> np.ndarray:
    ref = np.max(mag)
    log_spec = 10.0 * np.log10(np.maximum(amin * amin, mag * mag))
    log_spec -= 10.0 * np.log10(np.maximum(amin * amin, ref * ref))
    return np.maximum(log_spec, log_spec.max() - top_db)


def extract_perceptual_loudness(audio, sample_rate=16000, n_fft=2048, hop_length=512, window="hann", epsilon=1e-5,
                                interpolate_fn=None, normalise=True):
    if window != "hann":
        raise ValueError("only the reference's hann window is restated")
    db = amplitude_to_db_refmax(stft_magnitude(audio, n_fft, hop_length), epsilon)
    loudness = db.mean(axis=0)
    if interpolate_fn:
        loudness = interpolate_fn(loudness, n_fft, hop_length, original_length=np.asarray(audio).size)
    if normalise:
        loudness = (loudness + 80.0) / 80.0
    return loudness


def linear_interpolation(signal, window_length, hop_length, original_length=None):
    """data/utils/upsampling.py:20-35"""
    frames = signal.size
    padded = frames * hop_length + window_length - hop_length
    out = np.interp(np.linspace(0, frames - 1, padded), np.linspace(0, frames - 1, frames), signal)
    if original_length:
        out = out[window_length // 2:][:original_length]
    return out
