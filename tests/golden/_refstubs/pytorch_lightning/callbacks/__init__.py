from .model_checkpoint import ModelCheckpoint  # noqa: F401
