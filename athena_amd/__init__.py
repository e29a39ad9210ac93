"""athena_amd: MI355X-native implementation of Athena's hybrid attention/CTC ASR training step.

The public names mirror the reference's ``athena/__init__.py:50-87`` for this path.
"""
from .loss import CTCLoss, MPCLoss, Seq2SeqSparseCategoricalCrossentropy
from .metrics import Accuracy, CharactorAccuracy, CTCAccuracy, Seq2SeqSparseCategoricalAccuracy
from .models.base import BaseModel
from .models.masked_pc import MaskedPredictCoding
from .models.deep_speech import DeepSpeechModel
from .models.mtl_seq2seq import MtlTransformerCtc
from .models.speech_transformer import SpeechTransformer
from .solver import BaseSolver, DecoderSolver, HorovodSolver
from .tools.beam_search import BeamSearchDecoder
from .tools.ctc_scorer import CTCPrefixScorer
from .utils.checkpoint import Checkpoint
from .utils.hparam import HParams, register_and_parse_hparams
from .utils.learning_rate import ExponentialDecayAdam, WarmUpAdam, WarmUpLearningSchedule
from .utils.metric_check import MetricChecker, set_default_summary_writer
from .utils.misc import (create_multihead_mask, generate_square_subsequent_mask, insert_eos_in_labels,
                         insert_sos_in_labels, validate_seqs)

__all__ = [
    "CTCLoss", "MPCLoss", "MaskedPredictCoding", "DeepSpeechModel", "Seq2SeqSparseCategoricalCrossentropy", "Accuracy", "CharactorAccuracy", "CTCAccuracy",
    "Seq2SeqSparseCategoricalAccuracy", "BaseModel", "MtlTransformerCtc", "SpeechTransformer", "BaseSolver",
    "HorovodSolver", "DecoderSolver", "BeamSearchDecoder", "CTCPrefixScorer", "HParams", "register_and_parse_hparams", "ExponentialDecayAdam", "WarmUpAdam",
    "WarmUpLearningSchedule", "MetricChecker", "set_default_summary_writer", "create_multihead_mask",
    "generate_square_subsequent_mask", "insert_eos_in_labels", "insert_sos_in_labels", "validate_seqs",
    "Checkpoint",
]
