"""Frame-rate -> sample-rate interpolation of extracted features (mirror of the reference's
neural_waveshaping_synthesis/data/utils/upsampling.py:11-90: the three gin-selectable `interpolate_fn`s of the loudness / F0
features; host-side numpy / scipy like the reference: they post-process a few hundred frames per file and are not part of the
synthesis path).  Pinned on vectors recorded from the reference (tests/golden/g9_upsampling.npz)."""
from typing import Optional

import numpy as np

from ... import ginlite as gin


def get_padded_length(frames: int, window_length: int, hop_length: int) -> int:
    # the sample span `frames` centred analysis windows cover
    return (frames - 1) * hop_length + window_length


@gin.configurable
def linear_interpolation(signal: np.ndarray, window_length: int, hop_length: int, original_length: Optional[int] = None):
    """Piecewise-linear resampling of `signal` (one value per frame) onto the padded sample axis, then - when
    `original_length` is given - removal of the half window of centre padding and truncation to the audio length."""
    frames = signal.size
    frame_axis = np.linspace(0, frames - 1, frames)
    sample_axis = np.linspace(0, frames - 1, get_padded_length(frames, window_length, hop_length))
    out = np.interp(sample_axis, frame_axis, signal)
    if original_length:
        out = out[window_length // 2:][:original_length]
    return out


def _trim(out: np.ndarray, skip: int, original_length: Optional[int]):
    return out[skip:][:original_length] if original_length else out


@gin.configurable
def cubic_spline_interpolation(signal: np.ndarray, window_length: int, hop_length: int, original_length: Optional[int] = None):
    """The same resampling through the C2 cubic spline that interpolates every frame value (not-a-knot ends: what
    `scipy.interpolate.interp1d(kind="cubic")` of upsampling.py:44 builds), then the same trimming as the linear form."""
    import scipy.interpolate

    frames = signal.size
    spline = scipy.interpolate.make_interp_spline(np.arange(frames, dtype=np.float64), np.asarray(signal, dtype=np.float64), k=3)
    sample_axis = np.linspace(0, frames - 1, get_padded_length(frames, window_length, hop_length))
    return _trim(spline(sample_axis), window_length // 2, original_length)


@gin.configurable
def overlap_add_upsample(signal: np.ndarray, window_length: int, hop_length: int, window_fn: str = "hann", window_scale: int = 2,
                         original_length: Optional[int] = None):
    """Every frame value scales one `window_fn` window of `window_scale` hops placed at its hop (upsampling.py:56-83): the sum of
    those windows, i.e. the zero-stuffed frame sequence convolved with the window; with `original_length` the result is cut
    symmetrically out of the padded span."""
    import scipy.signal.windows

    frames = signal.size
    window = scipy.signal.windows.get_window(window_fn, hop_length * window_scale)
    padded = get_padded_length(frames, window_length, hop_length)
    out = np.zeros(padded)
    for start, value in zip(range(0, frames * hop_length, hop_length), np.asarray(signal, dtype=np.float64)):
        seg = out[start:start + window.size]          # (windows that reach past the padded span are cut there)
        seg += value * window[:seg.size]
    return _trim(out, (padded - original_length) // 2 if original_length else 0, original_length)
