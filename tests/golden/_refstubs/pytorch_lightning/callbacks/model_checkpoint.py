class ModelCheckpoint:
    """Exists only so the reference's .ckpt pickles resolve their one non-torch global."""
