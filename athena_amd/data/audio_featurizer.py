"""Acoustic front end of the data pipeline: log mel filterbank features from 16-bit PCM audio
(the default ``{"type": "Fbank"}`` of the reference's athena/transform/audio_featurizer.py:26-110,
computed there by compiled TensorFlow ops: athena/transform/feats/ops/kernels/spectrum.cc:76-172,
support_functions.cc:86-126,607-617, mfcc_mel_filterbank.cc:43-183, fbank.cc:42-70).

numpy on the host.  The arithmetic below follows those sources step by step -- frames of
int(0.025 sr) samples every int(0.010 sr), mean removed, pre-emphasis 0.97 inside the frame, the
"povey" window, power spectrum of the next power of two, TensorFlow's triangular mel bands between
``lower_frequency_limit`` and ``upper_frequency_limit``, natural log with floor 1e-12 -- but no value
produced by the reference's ops is available to check it against: the reference's unit test reads a
wav that is not part of the repository.  Dither (off by default) is not built.

Besides wav files (``scipy.io.wavfile``) the featurizer takes ``.npy`` files: a 1-D array is audio
at ``sample_rate``, a 2-D or 3-D array is a feature matrix [T, dim(, channels)] used as is.
"""
import numpy as np

from ..utils.hparam import register_and_parse_hparams


def _povey(n):
    k = np.arange(n, dtype=np.float64)
    return np.power(0.5 - 0.5 * np.cos(2.0 * np.pi * k / (n - 1)), 0.85).astype(np.float32)


def _mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_filterbank(num_bins, sample_rate, channels, low, high):
    """[num_bins, channels] weights of TensorFlow's MfccMelFilterbank: band edges equally spaced in
    mel, each FFT bin between the limits split between the band it falls into and the next one."""
    edges = _mel(low) + (_mel(high) - _mel(low)) / (channels + 1) * np.arange(1, channels + 2)
    hz_per_bin = 0.5 * sample_rate / (num_bins - 1)
    start, end = int(1 + low / hz_per_bin), int(high / hz_per_bin)
    w = np.zeros((num_bins, channels), dtype=np.float64)
    ch = 0
    for i in range(start, min(end, num_bins - 1) + 1):
        m = _mel(i * hz_per_bin)
        while ch < channels and edges[ch] < m:
            ch += 1
        lower_edge = edges[ch - 1] if ch >= 1 else _mel(low)
        upper_edge = edges[ch] if ch < len(edges) else edges[-1]
        down = (upper_edge - m) / (upper_edge - lower_edge)        # share of the band below
        if ch - 1 >= 0:
            w[i, ch - 1] += down
        if ch < channels:
            w[i, ch] += 1.0 - down
    return w


class AudioFeaturizer:
    default_config = {
        "type": "Fbank",
        "sample_rate": 16000,
        "window_length": 0.025,
        "frame_length": 0.010,
        "snip_edges": 1,
        "preEph_coeff": 0.97,
        "window_type": "povey",
        "remove_dc_offset": True,
        "upper_frequency_limit": 0,
        "lower_frequency_limit": 60,
        "filterbank_channel_count": 40,
        "is_log10": False,
        "delta_delta": False,
        "dither": 0.0,
    }

    def __init__(self, config=None):
        self.hparams = register_and_parse_hparams(self.default_config, config or {"type": "Fbank"}, cls=self.__class__)
        p = self.hparams
        if p.type != "Fbank":
            raise NotImplementedError("audio feature type %r is not built (only Fbank)" % (p.type,))
        if p.delta_delta or p.dither or p.window_type != "povey" or p.snip_edges != 1:
            raise NotImplementedError("Fbank: delta_delta, dither, other windows and snip_edges != 1 are not built")
        self.name = p.type
        self._banks = {}

    @property
    def dim(self):
        return self.hparams.filterbank_channel_count

    @property
    def num_channels(self):
        return 1

    def read(self, audio):
        """-> (float32 samples in the 16-bit range, sample rate) or (feature matrix, None)."""
        if isinstance(audio, str):
            if audio.endswith(".npy"):
                arr = np.load(audio, allow_pickle=False)
                if arr.ndim >= 2:
                    return arr.astype(np.float32), None
                return arr.astype(np.float32), self.hparams.sample_rate
            from scipy.io import wavfile
            sr, data = wavfile.read(audio)
            if data.ndim > 1:
                data = data[:, 0]
            if data.dtype.kind == "f":
                data = data * 32768.0
            return data.astype(np.float32), int(sr)
        arr = np.asarray(audio, dtype=np.float32)
        return arr.reshape(-1), self.hparams.sample_rate

    def __call__(self, audio=None, sr=None, speed=1.0):
        data, rate = self.read(audio)
        if rate is None:
            return data if data.ndim == 3 else data[:, :, None]
        rate = int(sr or rate)
        if speed != 1.0:
            # played ``speed`` times faster: the samples are re-read at that spacing (linear interpolation)
            n_out = int(len(data) / speed)
            data = np.interp(np.arange(n_out) * speed, np.arange(len(data)), data).astype(np.float32)
        return self.fbank(data, rate)

    def fbank(self, samples, rate):
        """[T, channels, 1] float32."""
        p = self.hparams
        win, hop = int(p.window_length * rate), int(p.frame_length * rate)
        n = max((len(samples) - win) // hop + 1, 1)
        nfft = 1 << int(np.ceil(np.log2(win)))
        idx = np.arange(n)[:, None] * hop + np.arange(win)[None, :]
        frames = np.where(idx < len(samples), samples[np.minimum(idx, len(samples) - 1)], 0.0).astype(np.float32)
        if p.remove_dc_offset:
            frames = frames - (frames.sum(axis=1, dtype=np.float32) / np.float32(win))[:, None]
        if p.preEph_coeff:
            coef = np.float32(p.preEph_coeff)
            shifted = np.concatenate([frames[:, :1], frames[:, :-1]], axis=1)
            frames = frames - coef * shifted
        spec = np.fft.rfft(frames * _povey(win)[None, :], n=nfft, axis=1)
        power = (spec.real.astype(np.float32) ** 2 + spec.imag.astype(np.float32) ** 2).astype(np.float64)
        high = p.upper_frequency_limit
        if high <= 0:
            high = rate / 2.0 + high
        elif high > rate / 2.0 or high <= p.lower_frequency_limit:
            high = rate / 2.0
        key = (power.shape[1], rate, p.filterbank_channel_count, p.lower_frequency_limit, high)
        if key not in self._banks:
            self._banks[key] = mel_filterbank(*key)
        mel = np.maximum(power @ self._banks[key], 1e-12)
        out = np.log10(mel) if p.is_log10 else np.log(mel)
        return out.astype(np.float32)[:, :, None]
