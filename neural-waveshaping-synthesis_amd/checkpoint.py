"""Checkpoint front end.

Reads either
  * a PyTorch-Lightning ``.ckpt`` as shipped by the reference (checkpoints/nws/*/last.ckpt: torch
    zip-pickle with ``state_dict`` and ``hyper_parameters``; the pickle references one
    pytorch_lightning class, resolved here with a throw-away stand-in so Lightning is not needed), or
  * a flat ``.npz`` of state-dict arrays (tests/golden/weights_vn.npz).
and the ``data_mean.npy`` / ``data_std.npy`` normalisation statistics next to it.
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch

_HPARAM_KEYS = ("n_waveshapers", "control_hop", "sample_rate", "learning_rate", "lr_decay", "lr_decay_interval",
                "log_audio")


@contextlib.contextmanager
def _lightning_standin():
    names = ["pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.callbacks.model_checkpoint"]
    added = []
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
            added.append(n)
    mc = sys.modules[names[2]]
    if not hasattr(mc, "ModelCheckpoint"):
        mc.ModelCheckpoint = type("ModelCheckpoint", (), {})
    try:
        yield
    finally:
        for n in added:
            sys.modules.pop(n, None)


def read_checkpoint(path):
    """-> (state_dict as {key: tensor/array}, hyper-parameters dict)."""
    if str(path).endswith(".npz"):
        z = np.load(path)
        state = {k: z[k] for k in z.files if not k.startswith("__")}
        if "__hparams__" in z.files:        # fixtures of non-default gin configurations carry their constructor arguments
            import json

            return state, {k: v for k, v in json.loads(str(z["__hparams__"])).items() if k in _HPARAM_KEYS}
        return state, dict(n_waveshapers=64, control_hop=128, sample_rate=16000)
    with _lightning_standin():
        ck = torch.load(path, map_location="cpu", weights_only=False)
    hp = {k: v for k, v in dict(ck.get("hyper_parameters", {})).items() if k in _HPARAM_KEYS}
    return dict(ck["state_dict"]), hp


def load_normalisation(checkpoint_dir):
    """data_mean / data_std rows 0 (f0) and 1 (loudness) as float64 (reference: colab cell 6, 15)."""
    mean = np.load(os.path.join(checkpoint_dir, "data_mean.npy")).astype(np.float64).reshape(-1)
    std = np.load(os.path.join(checkpoint_dir, "data_std.npy")).astype(np.float64).reshape(-1)
    return mean, std


def make_control(f0_hz, loudness, mean, std):
    """control = stack((f0 - mean0)/std0, (loudness - mean1)/std1); F0 itself stays in Hz."""
    f0_hz = np.asarray(f0_hz, dtype=np.float64)
    loudness = np.asarray(loudness, dtype=np.float64)
    c = np.stack([(f0_hz - mean[0]) / std[0], (loudness - mean[1]) / std[1]], axis=-2)
    return torch.as_tensor(f0_hz, dtype=torch.float32).unsqueeze(-2), torch.as_tensor(c, dtype=torch.float32)
