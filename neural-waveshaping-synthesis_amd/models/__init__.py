from .neural_waveshaping import ControlModule, NeuralWaveshaping  # noqa: F401
