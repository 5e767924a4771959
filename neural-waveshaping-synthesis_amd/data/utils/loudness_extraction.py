"""Perceptual-loudness feature on the MI355X (mirror of neural_waveshaping_synthesis/data/utils/loudness_extraction.py).

`extract_perceptual_loudness` keeps the reference's signature, defaults and gin binding names (:41-67), including
`interpolate_fn=linear_interpolation` (a plain call returns loudness at SAMPLE rate, length `audio.size`; the shipped
gin/data/urmp_4second_crepe.gin binds it to None = frame rate); the STFT / dB / mean chain runs in `csrc/loudness.hip`
(one windowed-DFT GEMM on the matrix cores + a dB pass).  Accepts a 1-D numpy array like the reference (returns numpy),
or a (N,) / (B, N) CUDA tensor (returns a tensor per row).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np
import torch

from ... import _lib
from ... import ginlite as gin
from ..._lib import check, ptr
from ...engine import ops, stream_ptr
from .upsampling import linear_interpolation

_DFT_CACHE: dict = {}


def _dft_operand(n_fft: int, device) -> torch.Tensor:
    key = (n_fft, str(device))
    t = _DFT_CACHE.get(key)
    if t is None:
        nbytes = _lib.lib().nws_loudness_dft_bytes(n_fft)
        if nbytes == 0:
            raise RuntimeError(f"n_fft must be a power of two in [64, 2048], got {n_fft}")
        with torch.cuda.device(device):
            t = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
            check(_lib.lib().nws_loudness_dft_matrix(n_fft, ptr(t), stream_ptr(device)), "nws_loudness_dft_matrix")
            torch.cuda.current_stream(device).synchronize()     # shared by every later caller, whatever its stream
        _DFT_CACHE[key] = t
    return t


def loudness_frames(audio: torch.Tensor, n_fft: int, hop_length: int, epsilon: float = 1e-5, top_db: float = 80.0,
                    normalise: bool = True) -> torch.Tensor:
    """(B, N) fp32 CUDA tensor -> (B, 1 + N // hop_length) loudness per frame."""
    if not audio.is_cuda or audio.dtype != torch.float32 or audio.dim() != 2:
        raise RuntimeError("audio: expected a (B, N) float32 CUDA tensor (no CPU fallback)")
    audio = audio.contiguous()
    B, N = audio.shape
    L = _lib.lib()
    frames = L.nws_loudness_frames(N, hop_length)
    nbytes = L.nws_loudness_workspace_bytes(B, N, n_fft, hop_length)
    if nbytes == 0 or N <= n_fft // 2:
        raise RuntimeError(f"unsupported loudness configuration: N={N}, n_fft={n_fft}, hop_length={hop_length} (n_fft: a power "
                           "of two in [64, 2048]; 1 <= hop_length <= n_fft; the 31 * hop_length + n_fft samples one workgroup "
                           "stages must fit 160 KB of LDS, e.g. hop_length <= 1250 at n_fft 2048; N > n_fft / 2)")
    dft = _dft_operand(n_fft, audio.device)
    o = ops()
    if o is not None:
        return o.loudness(audio, dft, int(n_fft), int(hop_length), float(epsilon), float(top_db), bool(normalise))
    with torch.cuda.device(audio.device):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=audio.device)
        out = torch.empty((B, frames), dtype=torch.float32, device=audio.device)
        check(L.nws_loudness(ptr(audio), B, N, n_fft, hop_length, ptr(dft), float(epsilon), float(top_db), 1 if normalise else 0,
                             ptr(out), ptr(ws), ws.numel(), stream_ptr(audio.device)), "nws_loudness")
    return out


@gin.configurable
def extract_perceptual_loudness(audio, sample_rate: float = 16000, n_fft: int = 2048, hop_length: int = 512,
                                window: str = "hann", epsilon: float = 1e-5,
                                interpolate_fn: Optional[Callable] = linear_interpolation, normalise: bool = True):
    """loudness_extraction.py:41-67.  (sample_rate only fed the A-weighting, which the reference computes and then does not
    apply, :38.)  interpolate_fn, if given, is called exactly like the reference calls it, on the host."""
    if window != "hann":
        raise RuntimeError("only the reference's hann window is implemented")
    is_numpy = isinstance(audio, np.ndarray)
    x = torch.as_tensor(np.ascontiguousarray(audio, dtype=np.float32)).cuda() if is_numpy else audio
    squeeze = x.dim() == 1
    if squeeze:
        x = x.unsqueeze(0)
    # normalisation is applied after the optional interpolation, as in the reference (both are affine, order kept anyway)
    out = loudness_frames(x, n_fft, hop_length, epsilon, 80.0, normalise=normalise and not interpolate_fn)
    if interpolate_fn:
        rows = [interpolate_fn(r, n_fft, hop_length, original_length=x.shape[1]) for r in out.cpu().numpy().astype(np.float64)]
        res = np.stack(rows)
        if normalise:
            res = (res + 80) / 80
        res = res[0] if squeeze else res
        return res if is_numpy else torch.as_tensor(res)
    if squeeze:
        out = out[0]
    return out.cpu().numpy() if is_numpy else out
