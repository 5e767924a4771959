from .dynamic import FiLM, TimeDistributedLayerNorm, TimeDistributedMLP  # noqa: F401
from .generators import FIRNoiseSynth, HarmonicOscillator  # noqa: F401
from .shaping import NEWT, FastNEWT, Reverb, Sine, TrainableNonlinearity  # noqa: F401
