"""Data side of the training step: the CSV-indexed speech-recognition data set with its text / audio
featurizers and CMVN, the synthetic dataset builder and fixed-shape synthetic batches."""
from .audio_featurizer import AudioFeaturizer  # noqa: F401
from .feature_normalizer import FeatureNormalizer  # noqa: F401
from .speech_recognition import BaseDatasetBuilder, SpeechRecognitionDatasetBuilder  # noqa: F401
from .synthetic import (DataDescriptions, SyntheticSpeechRecognitionDatasetBuilder, make_samples,  # noqa: F401
                        model_config)
from .text_featurizer import TextFeaturizer, Vocabulary  # noqa: F401
