"""MI355X-native kernel-prediction denoiser: the conv hot path of DeepBlender/DeepDenoiser."""
__version__ = "0.1.0"
