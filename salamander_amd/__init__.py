"""MI355X-native KL-NMF update engine behind Salamander's KLNMF / MvNMF API."""

__version__ = "0.1.0"
