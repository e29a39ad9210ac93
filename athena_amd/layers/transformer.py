"""Post-LN Transformer encoder/decoder (mirror of the reference's athena/layers/transformer.py:27-388)."""
import torch
from torch import nn

from ..runtime import truncated_normal
from .attention import AttnMask, MultiHeadAttention
from .commons import FFN_ACTIVATIONS, Dense, LayerNormParams


def _ffn_activation(name):
    if name not in FFN_ACTIVATIONS:
        raise NotImplementedError("ffn activation %r (built: %s)" % (name, ", ".join(sorted(FFN_ACTIVATIONS))))
    return FFN_ACTIVATIONS[name]


class TransformerEncoderLayer(nn.Module):
    """self-attention + FFN, each followed by dropout, residual add and LayerNorm(eps=1e-8)
    (transformer.py:237-310)."""

    def __init__(self, store, name, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="gelu",
                 unidirectional=False, look_ahead=0, conv_module_kernel_size=0):
        super().__init__()
        self.ffn_act, self.ffn_dact = _ffn_activation(activation)
        if conv_module_kernel_size > 0:
            raise NotImplementedError("conv_module_kernel_size > 0 (conformer conv module) is not built")
        self.store, self.rate = store, dropout
        self.self_attn = MultiHeadAttention(store, name + ".self_attn", d_model, nhead, unidirectional, look_ahead)
        init = truncated_normal(0.02)
        # the one-launch layer tail is built for the default gelu; any other activation runs the block
        # as separate products (no fragment-tiled weight copies, which is what selects the tail)
        tail = activation == "gelu"
        self.ffn0 = Dense(store, name + ".ffn.0", d_model, dim_feedforward, init, tiled=tail)
        self.ffn1 = Dense(store, name + ".ffn.1", dim_feedforward, d_model, init, tiled=tail)
        self.norm1 = LayerNormParams(store, name + ".norm1", d_model)
        self.norm2 = LayerNormParams(store, name + ".norm2", d_model)
        self.site_res1, self.site_ffn_h, self.site_res2 = store.new_site(), store.new_site(), store.new_site()

    def forward(self, src, src_mask=None, training=None):
        from .fused import EncoderLayerFn
        m = AttnMask.coerce(src_mask)
        return EncoderLayerFn.apply(src, self, m.key_mask, bool(training), *self.parameters())


class TransformerDecoderLayer(nn.Module):
    """masked self-attention, encoder-decoder attention, FFN (transformer.py:318-388)."""

    def __init__(self, store, name, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="gelu"):
        super().__init__()
        self.ffn_act, self.ffn_dact = _ffn_activation(activation)
        self.store, self.rate = store, dropout
        self.attn1 = MultiHeadAttention(store, name + ".attn1", d_model, nhead)
        self.attn2 = MultiHeadAttention(store, name + ".attn2", d_model, nhead)
        init = truncated_normal(0.02)
        tail = activation == "gelu"
        self.ffn0 = Dense(store, name + ".ffn.0", d_model, dim_feedforward, init, tiled=tail)
        self.ffn1 = Dense(store, name + ".ffn.1", dim_feedforward, d_model, init, tiled=tail)
        self.norm1 = LayerNormParams(store, name + ".norm1", d_model)
        self.norm2 = LayerNormParams(store, name + ".norm2", d_model)
        self.norm3 = LayerNormParams(store, name + ".norm3", d_model)
        self.site_res1, self.site_res2 = store.new_site(), store.new_site()
        self.site_ffn_h, self.site_res3 = store.new_site(), store.new_site()

    def cross_attention_weights(self, tgt, memory, tgt_mask=None, memory_mask=None):
        """The weights of this layer's encoder-decoder attention for the input ``tgt`` (the second
        value of TransformerDecoderLayer.call, transformer.py:383-388), without dropout: the
        self-attention sub-layer is run again eagerly to get the queries."""
        from .fused import LN_EPS
        from .. import ops
        with torch.no_grad():
            a, _ = self.attn1(tgt, tgt, tgt, tgt_mask)
            z, y, _, _ = ops.add_layernorm_fwd(tgt.contiguous(), a.contiguous(), self.norm1.gamma, self.norm1.beta,
                                               LN_EPS, 0.0, self.store.rng_state, 0)
            return self.attn2.attention_weights(memory, y, memory_mask)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, training=None):
        """Returns (output, None); the cross-attention weights are available through
        ``cross_attention_weights``."""
        from .fused import DecoderLayerFn
        tm, mm = AttnMask.coerce(tgt_mask), AttnMask.coerce(memory_mask)
        if not tm.causal and tgt_mask is not None:
            raise NotImplementedError("decoder self-attention expects the causal target mask")
        out = DecoderLayerFn.apply(tgt, memory, self, tm.key_mask, mm.key_mask, bool(training), *self.parameters())
        return out, None


class TransformerEncoder(nn.Module):
    """stack of encoder layers (transformer.py:149-185)."""

    def __init__(self, encoder_layers):
        super().__init__()
        self.layers = nn.ModuleList(encoder_layers)
        # a layer's one-launch tail can run the q / k / v projections of the layer behind it (a list
        # around the reference keeps the layer from being registered as a sub-module twice)
        for a, b in zip(encoder_layers[:-1], encoder_layers[1:]):
            if hasattr(a, "self_attn") and hasattr(b, "self_attn"):
                a.__dict__["_next_in_stack"] = [b]
                b.__dict__["_prev_in_stack"] = [a]     # backward: b hands its last product to a's tail

    def forward(self, src, src_mask=None, training=None):
        output = src
        n = len(self.layers)
        store = getattr(self.layers[0], "store", None) if n else None
        for layer in self.layers:
            layer.__dict__.pop("_pre_qkv", None)          # nothing left over from a pass that did not finish
            layer.__dict__.pop("_pre_x32", None)
            layer.__dict__.pop("_pending_dx", None)
        for i, layer in enumerate(self.layers):
            output = layer(output, src_mask=src_mask, training=training)
            # cut points for a pipelined backward pass: after every third of the stack
            if store is not None and store.collect_cuts and n >= 3 and i + 1 < n and (i + 1) % ((n + 2) // 3) == 0:
                store.cuts.append(("encoder.%d" % (i + 1), output))
        return output


class TransformerDecoder(nn.Module):
    """stack of decoder layers (transformer.py:188-234)."""

    def __init__(self, decoder_layers):
        super().__init__()
        self.layers = nn.ModuleList(decoder_layers)
        # as in the encoder stack: the last tail of a layer can run the self-attention projections of
        # the layer behind it
        for a, b in zip(decoder_layers[:-1], decoder_layers[1:]):
            if hasattr(a, "attn1") and hasattr(b, "attn1"):
                a.__dict__["_next_in_stack"] = [b]
                b.__dict__["_prev_in_stack"] = [a]
        # the gradient of the memory through all layers' cross-attention key / value projections as one
        # product: their kernels get one more bf16 copy, side by side
        cross = [l for l in decoder_layers if hasattr(l, "attn2")]
        if len(cross) > 1 and len({l.attn2.d_model for l in cross}) == 1 and cross[0].attn2.d_model % 8 == 0 and \
                not cross[0].store.finalized:
            cross[0].store.stack_shadows("decoder.memory_kv", [n for l in cross for n in
                                                              (l.attn2.wk.vname + ".kernel", l.attn2.wv.vname + ".kernel")])
            for i, l in enumerate(cross):
                l.__dict__["_kv_stack"] = (i, len(cross))
                l.__dict__["_kv_peers"] = [c.attn2 for c in cross]      # in the order of the stack

    def _project_memory(self, memory):
        """The key / value projections of the memory for ALL layers' cross-attention as one grouped
        launch (they depend on nothing a layer computes); each layer finds its pair waiting."""
        import os
        from . import fused
        from .. import ops
        layers = [l for l in self.layers if hasattr(l, "attn2")]
        if not layers or memory.dtype != torch.bfloat16 or 2 * len(layers) > 12 or \
                os.environ.get("ATHENA_TAIL_QKV", "1") == "0" or \
                os.environ.get("ATHENA_LAYER_TAIL_DECODER", "1") == "0":
            return
        B, T, d = memory.shape
        m2 = memory.reshape(B * T, d)
        if not all(ops.layer_tail_ok(m2, d, l.attn1.dense) and ops.layer_tail_ok(m2, d, l.attn2.dense, l.ffn0, l.ffn1)
                   for l in layers):
            return
        layers[0].store.refresh_shadows()
        outs = fused.lin_fwd_group([(m2, B * T, w) for l in layers for w in (l.attn2.wk, l.attn2.wv)])
        for i, l in enumerate(layers):
            l.__dict__["_pre_kv"] = (m2.data_ptr(), (B * T, d), (outs[2 * i], outs[2 * i + 1]))

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, return_attention_weights=False, training=None):
        output = tgt
        for layer in self.layers:
            layer.__dict__.pop("_pre_qkv", None)
            layer.__dict__.pop("_pre_x32", None)
            layer.__dict__.pop("_pending_dx", None)
            layer.__dict__.pop("_pre_kv", None)
        self._project_memory(memory)
        weights = []
        for layer in self.layers:
            if return_attention_weights:
                # the cross-attention weights of each layer (transformer.py:225-231), from the layer's input
                # to that attention: a separate pass, the fused layer kernels do not write them
                weights.append(layer.cross_attention_weights(output, memory, tgt_mask, memory_mask))
            output, _ = layer(output, memory, tgt_mask=tgt_mask, memory_mask=memory_mask, training=training)
        if return_attention_weights:
            return output, weights
        return output


class Transformer(nn.Module):
    """encoder + decoder (transformer.py:27-146)."""

    def __init__(self, store, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                 dim_feedforward=2048, dropout=0.1, activation="gelu", unidirectional=False, look_ahead=0,
                 custom_encoder=None, custom_decoder=None, conv_module_kernel_size=0):
        super().__init__()
        if custom_encoder is not None:
            self.encoder = custom_encoder
        else:
            self.encoder = TransformerEncoder([
                TransformerEncoderLayer(store, "encoder.%d" % i, d_model, nhead, dim_feedforward, dropout,
                                        activation, unidirectional, look_ahead, conv_module_kernel_size)
                for i in range(num_encoder_layers)])
        if custom_decoder is not None:
            self.decoder = custom_decoder
        else:
            self.decoder = TransformerDecoder([
                TransformerDecoderLayer(store, "decoder.%d" % i, d_model, nhead, dim_feedforward, dropout,
                                        activation)
                for i in range(num_decoder_layers)])
        self.d_model, self.nhead = d_model, nhead

    def forward(self, src, tgt, src_mask=None, tgt_mask=None, memory_mask=None, return_encoder_output=False,
                return_attention_weights=False, training=None):
        if src.shape[0] != tgt.shape[0]:
            raise RuntimeError("the batch number of src and tgt must be equal")
        if src.shape[2] != self.d_model or tgt.shape[2] != self.d_model:
            raise RuntimeError("the feature number of src and tgt must be equal to d_model")
        memory = self.encoder(src, src_mask=src_mask, training=training)
        output = self.decoder(tgt, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                              return_attention_weights=return_attention_weights, training=training)
        if return_encoder_output:
            return output, memory
        return output
