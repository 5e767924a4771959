"""Frame-rate -> sample-rate interpolation of extracted features (mirror of the reference's
neural_waveshaping_synthesis/data/utils/upsampling.py:11-36; host-side numpy like the reference: it post-processes a
few hundred frames per file and is not part of the synthesis path)."""
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
