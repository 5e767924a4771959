"""Plumbing for sub-modules called on their own (model.osc(f0), model.newt(exciter, emb), model.h_generator(emb), ...).

Inside NeuralWaveshaping.forward these modules never run as separate launches (everything is fused into five kernels);
called stand-alone each one maps to ONE stage kernel through torch.ops.newt_hip.* (or the ctypes binding of the same
C-ABI entry point with NWS_BACKEND=ctypes).  No PyTorch arithmetic fallback: CPU tensors raise."""
from __future__ import annotations

import ctypes as C

import torch

from ... import _lib
from ...engine import _req, ops, stream_ptr


def contiguous(t: torch.Tensor, name: str) -> torch.Tensor:
    return _req(t if t.is_contiguous() else t.contiguous(), name)


class Desc:
    """A partial NwsWeights descriptor for one sub-module, cached until one of its tensors changes."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, tensors: dict, scalars: dict | None = None):
        key = tuple((k, t.data_ptr(), t._version) for k, t in tensors.items()) + tuple(sorted((scalars or {}).items()))
        if key != self._key:
            no_autograd(params=list(tensors.values()))
            w = _lib.NwsWeights()
            keep = []
            for field, t in tensors.items():
                t = _req(t.detach(), field)
                keep.append(t)
                setattr(w, field, t.data_ptr())
            for field, v in (scalars or {}).items():
                setattr(w, field, v)
            self._val = (w, keep, torch.frombuffer(bytearray(bytes(w)), dtype=torch.uint8))
            self._key = key
        return self._val


def shaper_fields(sh) -> dict:
    """NwsWeights fields of a TrainableNonlinearity (reference shaping.py:15-37); the kernels are specialised for the
    architecture of gin/models/newt.gin"""
    if sh.depth != 4 or sh.width != 8 or sh.channels != 64:
        raise RuntimeError("kernels are specialised for 64 shapers, width 8, depth 4 (gin/models/newt.gin)")
    return {"shaper_in_scale": sh.input_scale, "shaper_w0": sh.net[0].weight, "shaper_b0": sh.net[0].bias,
            "shaper_w2": sh.net[2].weight, "shaper_b2": sh.net[2].bias, "shaper_w4": sh.net[4].weight,
            "shaper_b4": sh.net[4].bias, "shaper_w6": sh.net[6].weight, "shaper_b6": sh.net[6].bias}


_WARNED_PARAM_GRAD = [False]


def no_autograd(inputs=(), params=()):
    """The stage kernels are inference-only: the reference's modules are differentiable, these return tensors without a
    graph.  An INPUT that requires grad under grad mode means the caller expects gradients -> raise instead of silently
    returning a constant; parameters that require grad (the nn.Parameter default) only warn, once."""
    if not torch.is_grad_enabled():
        return
    for t in inputs:
        if isinstance(t, torch.Tensor) and t.requires_grad:
            raise RuntimeError("the HIP stage kernels are inference-only (no autograd): an input requires grad, so the result "
                               "would silently carry no graph.  Detach the input or call under torch.no_grad().")
    if not _WARNED_PARAM_GRAD[0] and any(isinstance(t, torch.Tensor) and t.requires_grad for t in params):
        import warnings

        _WARNED_PARAM_GRAD[0] = True
        warnings.warn("NEWT HIP kernels are inference-only: outputs carry no autograd graph although the module's parameters "
                      "require grad (wrap calls in torch.no_grad() to silence this)", stacklevel=3)


_ONES = {}


def sum_channels(x: torch.Tensor) -> torch.Tensor:
    """(B, C, N) -> (B, N): the `x.sum(1)` of models/neural_waveshaping.py:86 as a 1x1 convolution with unit weights
    (g_conv1x1_kernel: sequential adds in channel order)"""
    x = contiguous(x, "x")
    B, Cc, N = x.shape
    key = (x.device, Cc)
    ones = _ONES.get(key)
    if ones is None:
        ones = _ONES[key] = torch.ones((1, Cc), dtype=torch.float32, device=x.device)

    def c_call(L):
        with torch.cuda.device(x.device):
            y = torch.empty((B, 1, N), dtype=torch.float32, device=x.device)
            checked(L.nws_g_conv1x1(x.data_ptr(), ones.data_ptr(), None, B, Cc, 1, N, y.data_ptr(), stream_ptr(x.device)), "nws_g_conv1x1")
        return y

    return call("g_conv1x1", "nws_g_conv1x1", (x, ones, None), c_call)[:, 0].contiguous()


def has_hooks(*modules) -> bool:
    """forward (pre-)hooks registered on any of these modules"""
    return any(m is not None and (m._forward_hooks or m._forward_pre_hooks) for m in modules)


def call(op_name: str, c_name: str, op_args: tuple, c_call):
    """Run `torch.ops.newt_hip.<op_name>(*op_args)` or, on the ctypes binding, `c_call(lib)` (which returns the result)."""
    flat = []
    for a in op_args:
        flat.extend(a if isinstance(a, (list, tuple)) else (a,))
    no_autograd(inputs=flat)
    o = ops()
    if o is not None:
        return getattr(o, op_name)(*op_args)
    return c_call(_lib.lib())


def checked(rc: int, what: str):
    _lib.check(rc, what)


__all__ = ["C", "Desc", "call", "has_hooks", "no_autograd", "sum_channels", "checked", "contiguous", "shaper_fields", "stream_ptr", "_req", "_lib", "ops"]
