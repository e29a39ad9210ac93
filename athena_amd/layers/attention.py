"""Multi-head attention (mirror of the reference's athena/layers/attention.py:25-171)."""
import torch
from torch import nn

from ..runtime import truncated_normal
from .commons import Dense


class AttnMask:
    """Mask of an attention call: ``key_mask`` uint8 [B, Tk] (1 = masked key) and ``causal``
    (key j > query i masked; the kernels take it as a band width, see ``MultiHeadAttention.band``).  Equivalent to the reference's float masks [B,1,1,Tk] and
    [B,1,Tq,Tk] = max(key padding, look-ahead) (athena/utils/misc.py:68-90); the kernels add the
    reference's ``mask * -1e9`` to the scaled scores."""

    def __init__(self, key_mask, causal=False):
        self.key_mask, self.causal = key_mask, causal

    @staticmethod
    def coerce(mask):
        if mask is None:
            return AttnMask(None, False)
        if isinstance(mask, AttnMask):
            return mask
        if isinstance(mask, torch.Tensor) and mask.dim() == 4 and mask.shape[1] == 1 and mask.shape[2] == 1:
            return AttnMask((mask[:, 0, 0, :] > 0).to(torch.uint8).contiguous(), False)
        raise NotImplementedError("attention masks must be AttnMask objects or float tensors [B,1,1,Tk]")


class MultiHeadAttention(nn.Module):
    """wq / wk / wv / dense, all Dense(d_model) with truncated-normal(0.02) kernels
    (attention.py:98-137).  ``forward(v, k, q, mask)`` keeps the reference's argument order."""

    def __init__(self, store, name, d_model, num_heads, unidirectional=False, look_ahead=0):
        super().__init__()
        assert d_model % num_heads == 0
        if look_ahead < 0:
            raise ValueError("look_ahead must be >= 0")
        # band of a unidirectional layer (attention.py:55-58): query i sees keys j <= i + look_ahead;
        # stored as look_ahead + 1 so that 0 means "no band"
        self.band = int(look_ahead) + 1 if unidirectional else 0
        self.store, self.d_model, self.num_heads = store, d_model, num_heads
        self.depth = d_model // num_heads
        init = truncated_normal(0.02)
        # (tiled copies: the layer in front can run these three projections in its one-launch tail)
        self.wq = Dense(store, name + ".wq", d_model, d_model, init, tiled=True)
        self.wk = Dense(store, name + ".wk", d_model, d_model, init, tiled=True)
        self.wv = Dense(store, name + ".wv", d_model, d_model, init, tiled=True)
        self.dense = Dense(store, name + ".dense", d_model, d_model, init, tiled=True)
        # one fused bf16 copy [d, 3d] of the three projection kernels: the input gradients of the
        # projections that share an input are then a single product
        if d_model % 8 == 0:
            store.group_shadows([name + ".wq.kernel", name + ".wk.kernel", name + ".wv.kernel"])
        self._qkv_shadow = None

    def qkv_shadow(self):
        """Fused row-major bf16 copy [d, 3d] of (wq | wk | wv), or None in fp32 mode."""
        if self._qkv_shadow is None and self.store.shadow is not None and self.d_model % 8 == 0:
            self._qkv_shadow = self.store.group_shadow_views(self.wq.vname + ".kernel")[0]
        return self._qkv_shadow

    def attention_weights(self, k, q, mask):
        """softmax(q k^T / sqrt(depth) + mask) of this module's projections as an fp32 tensor
        [B, h, Tq, Tk]: the second value MultiHeadAttention.call returns in the reference
        (attention.py:161,171).  Computed on request, without gradient; the fused kernels of the
        forward pass never write it."""
        import math
        from .. import ops
        from .fused import lin_fwd
        m = AttnMask.coerce(mask)
        self.store.refresh_shadows()
        B, Tq, d = q.shape
        Tk = k.shape[1]
        with torch.no_grad():
            qp, _ = lin_fwd(q.contiguous().reshape(B * Tq, d), B * Tq, self.wq)
            kp, _ = lin_fwd(k.contiguous().reshape(B * Tk, d), B * Tk, self.wk)
            band = 1 if m.causal else self.band
            return ops.mha_weights(qp, kp, d, d, m.key_mask, band, B, self.num_heads, Tq, Tk, self.depth,
                                   1.0 / math.sqrt(self.depth))

    def forward(self, v, k, q, mask, return_attention_weights=False):
        """Returns (output, attention weights): the [B,h,Tq,Tk] weights are not materialised by the
        fused kernels and are None unless ``return_attention_weights`` asks for them (they then come
        from a separate pass, see ``attention_weights``)."""
        from .fused import MultiHeadAttentionFn
        if k is not v and not (k.data_ptr() == v.data_ptr() and k.shape == v.shape and k.stride() == v.stride()):
            raise NotImplementedError("MultiHeadAttention: k and v must be the same tensor (the transformer "
                                      "layers never pass different ones: transformer.py:304,383,385)")
        m = AttnMask.coerce(mask)
        self.store.refresh_shadows()
        # a look-ahead mask (band 1) and the layer's own band: the narrower one decides
        band = 1 if m.causal else self.band
        out = MultiHeadAttentionFn.apply(q.contiguous(), k.contiguous(), self, m.key_mask, band,
                                         *self.parameters())
        return out, (self.attention_weights(k, q, m) if return_attention_weights else None)
