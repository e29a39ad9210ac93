"""Text featurizers (mirror of the reference's athena/data/text_featurizer.py:27-240): character
vocabulary, space-separated English vocabulary, sentencepiece, and a word tokenizer fitted on the
transcripts themselves."""
import collections
import os
import re
import warnings

from ..utils.hparam import register_and_parse_hparams


class Vocabulary:
    """Character vocabulary read from a ``token index`` file; unknown tokens map to ``<unk>``."""

    def __init__(self, vocab_file):
        self.space, self.unk, self.eos = "<space>", "<unk>", "~"
        self.unk_index, self.max_index, self.eos_index = 0, 0, 0
        self.stoi, self.itos = {}, {}
        if vocab_file is not None:
            self.load_model(vocab_file)

    def load_model(self, vocab_file):
        if vocab_file is None or not os.path.exists(vocab_file):
            warnings.warn("[Warning] the vocab {} is not exists, make sure you are generating it, otherwise you "
                          "should check it!".format(vocab_file))
        self.stoi, self.itos = {}, {}
        self.unk_index, self.max_index, self.eos_index = 0, 0, 0
        with open(vocab_file, "r", encoding="utf-8") as vocab:
            for line in vocab:
                if line.startswith("#"):
                    continue
                word, index = line.split()
                index = int(index)
                self.itos[index] = word
                self.stoi[word] = index
                if word == self.unk:
                    self.unk_index = index
                if word == self.eos:
                    self.eos_index = index
                self.max_index = max(self.max_index, index)
        # a blank in a sentence is the <space> token
        if self.stoi.get(self.space, self.unk_index) != self.unk_index:
            self.stoi[" "] = self.stoi[self.space]
            self.itos[self.stoi[self.space]] = " "

    def __len__(self):
        return self.max_index + 1

    def _id(self, token):
        return self.stoi.get(token, self.unk_index)

    def decode(self, ids):
        return "".join(self.itos.get(i, self.unk) for i in ids)

    def encode(self, sentence):
        return [self._id(token.lower()) for token in list(sentence.strip())]

    def __call__(self, inputs):
        if isinstance(inputs, list):
            return self.decode(inputs)
        if isinstance(inputs, int):
            return self.itos.get(inputs, self.unk)
        if isinstance(inputs, str):
            return self.encode(inputs)
        raise ValueError("unsupported input")


class EnglishVocabulary(Vocabulary):
    """English vocabulary separated by space."""

    def decode(self, ids):
        return " ".join(self.itos.get(i, self.unk) for i in ids)

    def encode(self, sentence):
        return [self._id(token) for token in sentence.strip().split(" ")]


class SentencePieceFeaturizer:
    """sentencepiece model; sentences are upper-cased before encoding."""

    def __init__(self, spm_file):
        import sentencepiece as spm
        self.unk_index = 0
        self.sp = spm.SentencePieceProcessor()
        if spm_file is not None:
            self.sp.Load(spm_file)

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


class TextFeaturizer:
    """The main text featurizer interface."""
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
        self.punct_tokens = r"＇｛｝［］＼｜｀～＠＃＄％＾＆＊（）"
        self.punct_tokens += r"＿＋，。、‘’“”《》？：；【】——~！@"
        self.punct_tokens += r"￥%……&（）,.?<>:;\[\]|`\!@#$%^&()+?\"/_-"

    def load_model(self, model_file):
        self.model.load_model(model_file)

    @property
    def model_type(self):
        return self.p.type

    def delete_punct(self, tokens):
        return re.sub("[{}]".format(self.punct_tokens), "", tokens)

    def __len__(self):
        return len(self.model)

    def encode(self, texts):
        return self.model.encode(texts)

    def decode(self, sequences):
        return self.model.decode(sequences)

    @property
    def unk_index(self):
        if self.p.type == "vocab":
            return self.model.unk_index
        return -1
