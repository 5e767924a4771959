class FusedOnlyError(NotImplementedError):
    """Raised by sub-modules whose computation only exists fused inside the HIP forward."""


def fused_only(name: str, where: str):
    return FusedOnlyError(
        f"{name}.forward is not available stand-alone in the MI355X engine: it runs fused inside {where} "
        "(hand-written HIP kernels, no PyTorch fallback). Call NeuralWaveshaping.forward / render_exciter / "
        "get_embedding instead.")
