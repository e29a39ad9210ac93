"""Models of the hybrid attention/CTC training step (mirror of the reference's athena/models)."""
from .base import BaseModel
from .speech_transformer import SpeechTransformer
from .mtl_seq2seq import MtlTransformerCtc
from .masked_pc import MaskedPredictCoding
from .deep_speech import DeepSpeechModel
