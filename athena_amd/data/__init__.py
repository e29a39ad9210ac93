"""Data side of the training step: the synthetic dataset builder and fixed-shape synthetic batches."""
from .synthetic import (DataDescriptions, SyntheticSpeechRecognitionDatasetBuilder, make_samples,  # noqa: F401
                        model_config)
