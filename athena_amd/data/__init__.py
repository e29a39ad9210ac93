"""Dataset builders of the training entry point (reference athena/data/__init__.py)."""
from .synthetic import SyntheticSpeechRecognitionDatasetBuilder

__all__ = ["SyntheticSpeechRecognitionDatasetBuilder"]
