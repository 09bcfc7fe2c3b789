"""sais_amd — MI355X-native implementation of the SAIS ViT-over-video hot path."""
__version__ = "0.1.0"
