"""Decoding tools of the attention/CTC model (mirror of the reference's athena/tools for the
beam-search path: beam_search.py, ctc_scorer.py)."""
from .beam_search import BeamSearchDecoder, CandidateHolder
from .ctc_scorer import CTCPrefixScorer
