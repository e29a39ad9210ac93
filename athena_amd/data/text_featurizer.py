"""Transcript <-> label ids (behaviour of the reference's athena/data/text_featurizer.py:27-240).

Four text models behind one interface (``encode``, ``decode``, ``len``, ``load_model``):

* ``vocab``      a ``token index`` table, sentences split into lower-cased characters;
* ``eng_vocab``  the same table, sentences split on single blanks;
* ``spm``        a sentencepiece model, sentences upper-cased;
* ``text``       word ids by frequency rank over the transcripts the model is fitted on.

The two table models are ONE class (``IndexTable``) parameterised by a splitting rule; the public
names ``Vocabulary`` / ``EnglishVocabulary`` bind the rule.  ``TextFeaturizer`` picks the model
from the config (``{"type": ..., "model": ...}``) and adds punctuation removal.
"""
import collections
import os
import warnings

from ..utils.hparam import register_and_parse_hparams

SPACE_TOKEN, UNK_TOKEN, EOS_TOKEN = "<space>", "<unk>", "~"


class _Rule:
    """How a sentence becomes tokens and tokens become a sentence again."""

    def __init__(self, split, glue):
        self.split, self.glue = split, glue


CHARACTERS = _Rule(lambda s: [ch.lower() for ch in s.strip()], "")
BLANK_SEPARATED = _Rule(lambda s: s.strip().split(" "), " ")


def read_index_table(path):
    """``token index`` lines (``#`` lines skipped) -> (token -> id, id -> token).  A later line wins
    when a token or an id occurs twice, as with the dict assignments of the reference."""
    to_id, to_token = {}, {}
    with open(path, "r", encoding="utf-8") as f:
        for raw in f:
            if raw.startswith("#"):
                continue
            token, index = raw.split()
            to_id[token] = int(index)
            to_token[int(index)] = token
    return to_id, to_token


class IndexTable:
    """A token table with an out-of-vocabulary id.  Unknown tokens encode to the id of ``<unk>`` (0
    when the table has none); a table that knows ``<space>`` under another id than ``<unk>`` also
    encodes the blank character to it and decodes it back to a blank."""
    rule = CHARACTERS

    def __init__(self, vocab_file=None, rule=None):
        if rule is not None:
            self.rule = rule
        self.space, self.unk, self.eos = SPACE_TOKEN, UNK_TOKEN, EOS_TOKEN
        self._install({}, {})
        if vocab_file is not None:
            self.load_model(vocab_file)

    def _install(self, to_id, to_token):
        self.stoi, self.itos = to_id, to_token
        self.unk_index = to_id.get(UNK_TOKEN, 0)
        self.eos_index = to_id.get(EOS_TOKEN, 0)
        self.max_index = max(to_token, default=0)
        blank = to_id.get(SPACE_TOKEN, self.unk_index)
        if blank != self.unk_index:
            to_id[" "] = blank
            to_token[blank] = " "

    def load_model(self, vocab_file):
        if vocab_file is None or not os.path.exists(vocab_file):
            warnings.warn("[Warning] the vocab {} is not exists, make sure you are generating it, otherwise you "
                          "should check it!".format(vocab_file))
        self._install(*read_index_table(vocab_file))

    def __len__(self):
        return self.max_index + 1

    def encode(self, sentence):
        unk = self.unk_index
        return [self.stoi.get(tok, unk) for tok in self.rule.split(sentence)]

    def decode(self, ids):
        return self.rule.glue.join(self.itos.get(i, UNK_TOKEN) for i in ids)

    def __call__(self, inputs):
        """ids -> sentence, one id -> its token, sentence -> ids"""
        if isinstance(inputs, list):
            return self.decode(inputs)
        if isinstance(inputs, int):
            return self.itos.get(inputs, UNK_TOKEN)
        if isinstance(inputs, str):
            return self.encode(inputs)
        raise ValueError("unsupported input")


class Vocabulary(IndexTable):
    """Character vocabulary (reference ``Vocabulary``, text_featurizer.py:27-110)."""
    rule = CHARACTERS


class EnglishVocabulary(IndexTable):
    """Blank-separated English vocabulary (reference ``EnglishVocabulary``, text_featurizer.py:112-136)."""
    rule = BLANK_SEPARATED


class SentencePieceFeaturizer:
    """sentencepiece model; sentences are upper-cased before encoding (text_featurizer.py:138-176)."""
    unk_index = 0

    def __init__(self, spm_file=None):
        import sentencepiece
        self.sp = sentencepiece.SentencePieceProcessor()
        if spm_file is not None:
            self.load_model(spm_file)

    def load_model(self, model_file):
        self.sp.Load(model_file)

    def __len__(self):
        return self.sp.GetPieceSize()

    def encode(self, sentence):
        return self.sp.EncodeAsIds(sentence.upper())

    def decode(self, ids):
        return self.sp.DecodeIds(ids)


class TextTokenizer:
    """Word ids by frequency rank over the texts it is fitted on, the behaviour of
    tf.keras.preprocessing.text.Tokenizer with its defaults: lower-cased, punctuation
    ``!"#$%&()*+,-./:;<=>?@[\\]^_`{|}~`` plus tab and newline turned into blanks, split on blanks,
    ids from 1 in order of falling count (first seen first among equals); unseen words are skipped."""
    _FILTERS = '!"#$%&()*+,-./:;<=>?@[\\]^_`{|}~\t\n'

    def __init__(self, text=None):
        self.word_index = {}
        self.index_word = {}
        self.text = text
        if text is not None:
            self.load_model(text)

    def _words(self, text):
        text = text.lower().translate(str.maketrans({c: " " for c in self._FILTERS}))
        return [w for w in text.split(" ") if w]

    def load_model(self, text):
        counts = collections.OrderedDict()
        for sentence in text:
            for w in self._words(sentence):
                counts[w] = counts.get(w, 0) + 1
        ranked = sorted(counts.items(), key=lambda kv: -kv[1])          # stable: first seen first among equals
        self.word_index = {w: i + 1 for i, (w, _) in enumerate(ranked)}
        self.index_word = {i: w for w, i in self.word_index.items()}

    def __len__(self):
        return len(self.word_index) + 1

    def encode(self, texts):
        return [self.word_index[w] for w in self._words(texts) if w in self.word_index]

    def decode(self, sequences):
        return " ".join(self.index_word[i] for i in sequences if i in self.index_word)


# Characters ``delete_punct`` removes: the ASCII marks and their CJK / full-width counterparts that
# the character class of the reference (text_featurizer.py:198-200) matches, by code point.
_PUNCTUATION = frozenset(
    '!"#$%&()+,-./:;<>?@[]^_`|~'
    "—‘’“”…"                      # em dash, curly quotes, ellipsis
    "、。《》【】"                      # CJK comma, full stop, brackets
    "！＃＄％＆＇（）＊＋，：；？＠"
    "［＼］＾＿｀｛｜｝～￥")
_DROP_PUNCTUATION = {ord(c): None for c in _PUNCTUATION}


class TextFeaturizer:
    """The text model named by ``config["type"]``, built from ``config["model"]`` (a table file, a
    sentencepiece file, or -- for ``text`` -- nothing until ``load_model`` is given the transcripts)."""
    supported_model = {
        "vocab": Vocabulary,
        "eng_vocab": EnglishVocabulary,
        "spm": SentencePieceFeaturizer,
        "text": TextTokenizer,
    }
    default_config = {
        "type": "text",
        "model": None,
    }

    def __init__(self, config=None):
        self.p = register_and_parse_hparams(self.default_config, config)
        self.model = self.supported_model[self.p.type](self.p.model)

    @property
    def model_type(self):
        return self.p.type

    @property
    def unk_index(self):
        """Only the character table reports an out-of-vocabulary id; -1 says "this model has none"."""
        return self.model.unk_index if self.p.type == "vocab" else -1

    def load_model(self, model_file):
        self.model.load_model(model_file)

    def delete_punct(self, tokens):
        return tokens.translate(_DROP_PUNCTUATION)

    def __len__(self):
        return len(self.model)

    def encode(self, texts):
        return self.model.encode(texts)

    def decode(self, sequences):
        return self.model.decode(sequences)
