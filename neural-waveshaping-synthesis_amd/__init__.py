"""MI355X-native Neural Waveshaping Synthesis (NEWT) inference engine.

    import importlib
    nws = importlib.import_module("neural-waveshaping-synthesis_amd")   # or: import nws_amd as nws
    nws.gin.parse_config_file(nws.DEFAULT_GIN)
    model = nws.NeuralWaveshaping.load_from_checkpoint("last.ckpt").to("cuda")
    model.newt = nws.FastNEWT(model.newt)
    audio = model(f0, control)
"""
import os as _os

# ForwardPipeline runs 3+ streams beside the caller's own (and RCCL's); HIP folds streams onto 4 hardware queues by default
# and streams sharing a queue serialise.  Only effective if the HIP runtime has not started yet; harmless otherwise.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import ginlite as gin  # noqa: F401,E402
from ._lib import LIB_PATH, NwsError  # noqa: F401
from .models.neural_waveshaping import ControlModule, NeuralWaveshaping, ensure_default_config, _DEFAULT_GIN as DEFAULT_GIN  # noqa: F401,E501
from .models.modules.dynamic import FiLM, TimeDistributedLayerNorm, TimeDistributedMLP  # noqa: F401
from .models.modules.generators import FIRNoiseSynth, HarmonicOscillator  # noqa: F401
from .models.modules.shaping import NEWT, FastNEWT, Reverb, Sine, TrainableNonlinearity  # noqa: F401
from .pipeline import ForwardPipeline  # noqa: F401

__version__ = "0.1.0"
