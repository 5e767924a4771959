"""Stub of the tiny part of pytorch_lightning the reference's model file touches."""
import torch
import torch.nn as nn
from . import callbacks  # noqa: F401


class LightningModule(nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu", **kw):
        ck = torch.load(path, map_location=map_location, weights_only=False)
        model = cls(**ck.get("hyper_parameters", {}))
        model.load_state_dict(ck["state_dict"])
        return model


class LightningDataModule:
    pass


class Trainer:
    pass
