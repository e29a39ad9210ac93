"""Forward/backward sequences of the model's blocks as coarse ``torch.autograd.Function``s.

Each Function enqueues the HIP kernels of one block (x_net, y_net, an encoder layer, a decoder
layer, a Dense head) and keeps the intermediates it needs for the backward pass.  Parameter
gradients are written by the kernels straight into the flat gradient buffer (``param.grad`` is a
view of it) and ``None`` is returned to autograd for them, so a backward pass OVERWRITES
``.grad`` instead of accumulating into it.

Reference sequences: athena/layers/transformer.py:296-310 (encoder layer), :374-388 (decoder
layer), athena/layers/attention.py:147-171 (multi-head attention),
athena/models/speech_transformer.py:72-113 (x_net / y_net).
"""
import math
import os

import torch

from .. import ops
from ..ops import ACT_GELU, ACT_MUL_DGELU, ACT_NONE, ACT_RELU6

LN_EPS = 1e-8        # transformer.py:286-288
BN_EPS = 1e-3        # Keras BatchNormalization default
BN_MOMENTUM = 0.99   # Keras BatchNormalization default


# ----------------------------------------------------------------------------------------------
# Dense helpers.  ``dense`` is an athena_amd.layers.commons.Dense (kernel [K,N] fp32, bias [N]).
# ----------------------------------------------------------------------------------------------
def lin_fwd(x, M, dense, act=ACT_NONE, want_pre=False, drop_p=0.0, rng=None, site=0, out_dtype=None, ldc=None):
    """y[M,N] = act(x[M,K] @ kernel + bias) (+dropout); returns (y, pre-activation or None)."""
    K, N = dense.in_features, dense.out_features
    ldc = ldc or N
    out = torch.empty((M, ldc), dtype=out_dtype or x.dtype, device=x.device)
    pre = torch.empty_like(out) if want_pre else None
    if x.dtype == torch.bfloat16 and ops.PANEL_ENABLED and N == 256 and ldc == N and K % 256 == 0 and \
            512 <= K <= 256 * ops.PANEL_MAX_BLOCKS and getattr(dense, "tiled_t", None) is not None and \
            act in (ACT_NONE, ACT_RELU6) and not want_pre and drop_p == 0.0 and out.dtype == torch.bfloat16 and \
            x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 and M * x.stride(0) < 2 ** 31:
        # 256 outputs, a long contraction: the output rows stay in registers over all of it
        ops.panel_gemm_k(x, M, K, [(dense.tiled_t[(256 * c // 32) * 512:], K) for c in range(K // 256)], out, N,
                         bias=dense.bias, act=act)
        return out, None
    if x.dtype == torch.bfloat16:
        ops.gemm(x, dense.shadow_t, out, M, N, K, trans_b=True, lda=K, ldb=dense.shadow_t.shape[1], ldc=ldc,
                 bias=dense.bias, act=act, c_pre=pre, drop_p=drop_p, rng_state=rng, rng_site=site)
    else:
        ops.gemm(x, dense.kernel, out, M, N, K, trans_b=False, lda=K, ldb=N, ldc=ldc, bias=dense.bias, act=act,
                 c_pre=pre, drop_p=drop_p, rng_state=rng, rng_site=site)
    return out, pre


def lin_fwd_group(items):
    """y_i = x_i @ kernel_i + bias_i for several Dense layers at once (one grouped launch)."""
    x0, M0 = items[0][0], items[0][1]
    if ops.PANEL_ENABLED and x0.dtype == torch.bfloat16 and x0.is_contiguous() and x0.data_ptr() % 16 == 0 and \
            len(items) <= ops.PANEL_MAX_BLOCKS and \
            all(x.data_ptr() == x0.data_ptr() and x.shape == x0.shape and M == M0 and d.in_features == 256 and d.out_features == 256 and d.tiled_t is not None
                for x, M, d in items):
        # one input, 256 x 256 kernels with fragment-tiled copies: the rows stay in LDS for all of them
        outs = [torch.empty((M0, 256), dtype=x0.dtype, device=x0.device) for _ in items]
        ops.panel_gemm(x0.view(-1, 256), M0, [(d.tiled_t, d.bias, o, 256) for (_, _, d), o in zip(items, outs)])
        return outs
    outs, problems = [], []
    for x, M, dense in items:
        K, N = dense.in_features, dense.out_features
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
        outs.append(out)
        if x.dtype == torch.bfloat16:
            problems.append((x, dense.shadow_t, out, M, N, K, dict(trans_b=True, lda=K, ldb=dense.shadow_t.shape[1],
                                                                   ldc=N, bias=dense.bias)))
        else:
            problems.append((x, dense.kernel, out, M, N, K, dict(trans_b=False, lda=K, ldb=N, ldc=N,
                                                                 bias=dense.bias)))
    ops.gemm_grouped(problems)
    return outs


def lin_bwd_x(dy, M, dense, lddy, out=None, accumulate=False, act=ACT_NONE, aux=None, drop_p=0.0, rng=None,
              site=0):
    """dx[M,K] (+)= (dy[M,N] @ kernel^T) * act'(aux) * dropout."""
    K, N = dense.in_features, dense.out_features
    if out is None:
        out = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
    ldaux = aux.shape[-1] if aux is not None else 0
    if dy.dtype == torch.bfloat16:
        kc = ops.roundup(N, 8)    # padding columns of dy and of the shadow are zero
        if lddy < kc:
            raise RuntimeError("lin_bwd_x: dy leading dimension %d < %d" % (lddy, kc))
        # a plain product with few rows and a long contraction (the vocabulary projection of the
        # decoder: 1056 x 256 x 4240 is 17 workgroups) is left to split its contraction
        plain = act == ACT_NONE and aux is None and not accumulate and drop_p == 0.0
        if plain and ops.PANEL_ENABLED and N == 256 and K % 256 == 0 and K // 256 <= ops.PANEL_MAX_BLOCKS and \
                getattr(dense, "tiled", None) is not None and lddy % 8 == 0 and dy.data_ptr() % 16 == 0 and \
                out.data_ptr() % 16 == 0 and M * K < 2 ** 31:
            # 256 outputs, many inputs: dy's rows stay in LDS while the kernel's rows go by in groups of 256
            ops.panel_gemm(dy.view(-1, lddy) if dy.dim() != 2 else dy, M,
                           [(dense.tiled[65536 * c:], None, out.view(-1)[256 * c:], K) for c in range(K // 256)])
            return out
        ops.gemm(dy, dense.shadow, out, M, K, kc, trans_b=True, lda=lddy, ldb=dense.shadow.shape[1], ldc=K,
                 act=act, aux=aux, ldaux=ldaux, accumulate=accumulate, drop_p=drop_p, rng_state=rng, rng_site=site,
                 split_k=0 if (plain and kc >= 2048) else 1)
    else:
        ops.gemm(dy, dense.kernel, out, M, K, N, trans_b=True, lda=lddy, ldb=N, ldc=K, act=act, aux=aux,
                 ldaux=ldaux, accumulate=accumulate, drop_p=drop_p, rng_state=rng, rng_site=site)
    return out


class weight_grad_group:
    """Collects the weight-gradient products issued inside the ``with`` block and runs them as one
    grouped launch at its end: the weight gradients of a layer are independent of everything else
    in its backward pass, and together they fill the chip where each alone does not."""
    _open = None

    def __enter__(self):
        self.prev = weight_grad_group._open
        self.problems = []
        weight_grad_group._open = self
        return self

    def __exit__(self, exc_type, exc, tb):
        weight_grad_group._open = self.prev
        if exc_type is None:
            ops.gemm_grouped(self.problems)
        self.problems = None
        return False


class _WgradQueue:
    """Dense weight gradients of a whole backward pass, run as ONE launch at its end.

    A weight gradient x^T dy contracts over all rows of the batch and nothing later in the backward
    pass depends on it, so in bf16 mode ``lin_bwd_w`` only notes the operands here (which keeps them
    alive) and an end-of-backward callback of the autograd engine hands the whole list to
    ``ops.gemm_wgrad_table``: every 256x256 tile of every gradient is one workgroup that runs down its
    whole contraction -- no K split, no partial sums, one launch instead of one per layer.
    Outside a backward pass (no engine to call back) the product runs at once."""
    items = []
    queued = False
    enabled = os.environ.get("ATHENA_WGRAD_TABLE", "1") != "0"
    # A stream to run the table on instead of the launch stream (None: the launch stream).  The
    # data-parallel step cuts its backward pass into pieces and exchanges each piece's gradients on a
    # side stream while the next piece runs; a piece's table has too few tiles to fill the chip, so it
    # goes to that side stream as well -- in front of the exchange that needs its results, beside the
    # next piece's kernels -- instead of standing in the way on the launch stream.
    side = None

    @classmethod
    def push(cls, item):
        if not cls.queued:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(cls.flush)
            except RuntimeError:
                return False
            cls.queued = True
        cls.items.append(item)
        return True

    @classmethod
    def flush(cls):
        items, cls.items, cls.queued = cls.items, [], False
        if cls.side is None or not items:
            ops.gemm_wgrad_table(items)
            return
        main = torch.cuda.current_stream()
        cls.side.wait_stream(main)                      # the operands are complete on the launch stream
        for it in items:
            it[0].record_stream(cls.side)               # x and dy must outlive the side stream's use of them
            it[1].record_stream(cls.side)
        with torch.cuda.stream(cls.side):
            ops.gemm_wgrad_table(items)


def lin_bwd_w(x, dy, M, dense, ldx, lddy, defer=True):
    """kernel.grad = x^T @ dy ; bias.grad = column sums of dy.  ``defer=False`` runs the product now
    even inside a ``weight_grad_group`` -- needed when ``dy`` is about to be overwritten."""
    K, N = dense.in_features, dense.out_features
    aligned = ldx % 8 == 0 and lddy % 8 == 0 and ldx >= ops.roundup(K, 8) and lddy >= ops.roundup(N, 8) and \
        x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0
    if defer and aligned and _WgradQueue.enabled and x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and \
            _WgradQueue.push((x, dy, M, K, N, ldx, lddy, dense.kernel.grad, dense.bias.grad)):
        return
    kw = dict(trans_a=True, trans_b=False, lda=ldx, ldb=lddy, ldc=N, split_k=0, b_colsum=dense.bias.grad)
    grp = weight_grad_group._open
    if grp is not None and defer:
        grp.problems.append((x, dy, dense.kernel.grad, K, N, M, kw))   # keeps x and dy alive until the launch
    else:
        ops.gemm(x, dy, dense.kernel.grad, K, N, M, **kw)


# ----------------------------------------------------------------------------------------------
# Multi-head attention block (reference athena/layers/attention.py:147-171)
# ----------------------------------------------------------------------------------------------
def mha_block_fwd(q_in, kv_in, attn, B, Tq, Tk, key_mask, causal, project=True, qkv=None):
    """q/k/v projections (unless the caller brings them: ``qkv``, a triple whose missing members are
    None), attention and (``project``) the output projection.  Without it the attention context is
    returned in place of the block's output."""
    d, h = attn.d_model, attn.num_heads
    dk = d // h
    have = list(qkv) if qkv is not None else [None, None, None]
    todo = [(i, item) for i, item in enumerate([(q_in, B * Tq, attn.wq), (kv_in, B * Tk, attn.wk),
                                                (kv_in, B * Tk, attn.wv)]) if have[i] is None]
    if todo:
        for (i, _), out in zip(todo, lin_fwd_group([item for _, item in todo])):
            have[i] = out
    q, k, v = have
    ctx = torch.empty((B * Tq, d), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, h, Tq), dtype=torch.float32, device=q.device)
    ops.mha_fwd(q, k, v, d, d, d, key_mask, causal, B, h, Tq, Tk, dk, 1.0 / math.sqrt(dk), ctx, d, lse)
    if not project:
        return ctx, (q, k, v, ctx, lse)
    out, _ = lin_fwd(ctx, B * Tq, attn.dense)
    return out, (q, k, v, ctx, lse)


def mha_block_bwd(dout, q_in, kv_in, attn, saved, B, Tq, Tk, key_mask, causal, dq_in, acc_q, dkv_in, acc_kv,
                  dctx=None, hand_over=None, dkv_out=None):
    """Writes parameter grads; adds/stores the input grads into dq_in / dkv_in (may be one tensor).
    With ``dctx`` (the gradient of the attention context) the output projection is the caller's.
    ``hand_over``: a callable that takes the self-attention's dq | dk | dv buffer and sees to the
    product with the fused kernels itself (the input gradient is then short of it on return).
    ``dkv_out`` = (buffer [Mk, 2d] as a column block of a wider matrix, its leading dimension): the
    cross-attention's dk | dv go there and the product for dkv_in is the caller's."""
    d, h = attn.d_model, attn.num_heads
    dk = d // h
    q, k, v, ctx, lse = saved
    Mq, Mk = B * Tq, B * Tk
    if dctx is None:
        # without dropout the caller's dout IS the buffer the input gradients are accumulated into
        safe = dout.data_ptr() not in (dq_in.data_ptr(), dkv_in.data_ptr())
        lin_bwd_w(ctx, dout, Mq, attn.dense, d, d, defer=safe)
        dctx = lin_bwd_x(dout, Mq, attn.dense, d)
    delta = torch.empty_like(lse)
    fused = attn.qkv_shadow() if q.dtype == torch.bfloat16 else None
    same_in = q_in.data_ptr() == kv_in.data_ptr() and Mq == Mk and dq_in.data_ptr() == dkv_in.data_ptr()
    if fused is not None and same_in:
        # self-attention: dq | dk | dv side by side, so the gradient of the shared input is one
        # product with the fused [d, 3d] copy of the three kernels
        dqkv = torch.empty((Mq, 3 * d), dtype=q.dtype, device=q.device)
        dq, dk_, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
        ld = 3 * d
    elif fused is not None:
        # cross-attention: dk | dv side by side for the gradient of the memory
        dq = torch.empty_like(q)
        if dkv_out is not None:
            dkv, ld = dkv_out
        else:
            dkv, ld = torch.empty((Mk, 2 * d), dtype=k.dtype, device=k.device), 2 * d
        dk_, dv = dkv[:, :d], dkv[:, d:]
    else:
        dq = torch.empty_like(q)
        dk_ = torch.empty_like(k)
        dv = torch.empty_like(v)
        ld = d
    lddq = ld if (fused is not None and same_in) else d
    ops.mha_bwd(q, k, v, d, d, d, key_mask, causal, B, h, Tq, Tk, dk, 1.0 / math.sqrt(dk), ctx, dctx, d, lse,
                delta, dq, dk_, dv, lddq, ld, ld)
    lin_bwd_w(q_in, dq, Mq, attn.wq, d, lddq)
    lin_bwd_w(kv_in, dk_, Mk, attn.wk, d, ld)
    lin_bwd_w(kv_in, dv, Mk, attn.wv, d, ld)
    if fused is not None and same_in and hand_over is not None and acc_q:
        hand_over(dqkv)
    elif fused is not None and same_in:
        ops.gemm(dqkv, fused, dq_in, Mq, d, 3 * d, trans_b=True, lda=3 * d, ldb=3 * d, ldc=d, accumulate=acc_q)
    elif fused is not None:
        if hand_over is not None and acc_q:
            hand_over(dq)                       # cross-attention: the query side's product is the caller's
        else:
            lin_bwd_x(dq, Mq, attn.wq, d, out=dq_in, accumulate=acc_q)
        if dkv_out is None:
            ops.gemm(dkv, fused[:, d:], dkv_in, Mk, d, 2 * d, trans_b=True, lda=2 * d, ldb=3 * d, ldc=d,
                     accumulate=acc_kv)
    else:
        lin_bwd_x(dq, Mq, attn.wq, d, out=dq_in, accumulate=acc_q)
        lin_bwd_x(dk_, Mk, attn.wk, d, out=dkv_in, accumulate=acc_kv)
        lin_bwd_x(dv, Mk, attn.wv, d, out=dkv_in, accumulate=True)


def ffn_fwd(x, M, layer, p, rng):
    """Dense(dff, gelu) -> Dropout -> Dense(d); the trailing Dropout is applied by the caller's
    add+LayerNorm kernel (transformer.py:262-282)."""
    h, h_pre = lin_fwd(x, M, layer.ffn0, act=getattr(layer, "ffn_act", ACT_GELU), want_pre=True, drop_p=p, rng=rng,
                       site=layer.site_ffn_h)
    f, _ = lin_fwd(h, M, layer.ffn1)
    return f, (h_pre, h)


def ffn_bwd(df, x, M, layer, saved, p, rng, dx):
    """Parameter grads of the two Dense layers; dx += d(ffn)/dx."""
    h_pre, h = saved
    d, dff = layer.ffn0.in_features, layer.ffn0.out_features
    lin_bwd_w(h, df, M, layer.ffn1, dff, d, defer=df.data_ptr() != dx.data_ptr())
    dh = lin_bwd_x(df, M, layer.ffn1, d, act=getattr(layer, "ffn_dact", ACT_MUL_DGELU), aux=h_pre, drop_p=p, rng=rng,
                   site=layer.site_ffn_h)
    lin_bwd_w(x, dh, M, layer.ffn0, d, dff)
    lin_bwd_x(dh, M, layer.ffn0, dff, out=dx, accumulate=True)


def _params_of(*mods):
    ps = []
    for m in mods:
        ps.extend(m.parameters())
    return ps


def _add_pending(pend, dy):
    """dy += dqkv (wq | wk | wv)^T of the layer behind: the product that layer handed over, as a
    launch of its own."""
    dqkv, attn = pend[2], pend[3]
    M, d = dy.shape
    ops.gemm(dqkv, attn.qkv_shadow(), dy, M, d, 3 * d, trans_b=True, lda=3 * d, ldb=3 * d, ldc=d, accumulate=True)


def _encoder_tail_bwd(dy, t, x2, s_att, layer, B, T, key_mask, p, rng, pend=None, hand_over=None):
    """Backward pass of an encoder layer whose forward pass ran ``ops.layer_tail_fwd``.  ``pend``: what
    the layer behind handed over (see ``EncoderLayerFn.backward``); ``hand_over``: where this layer's
    own dq | dk | dv goes instead of into a product here."""
    M, d = dy.shape
    dff = layer.ffn0.out_features
    attn = layer.self_attn
    add = None
    if pend is not None:
        a2 = pend[3]
        add = (pend[2], (a2.wq, a2.wk, a2.wv))
    r = ops.layer_tail_bwd(dy, t, attn.dense, layer.norm1, layer.ffn0, layer.ffn1, layer.norm2, p, rng,
                           layer.site_res1, layer.site_ffn_h, layer.site_res2, add=add)
    if r is None and pend is not None:
        _add_pending(pend, dy)
    with weight_grad_group():
        if r is None:
            # no backward pass to fold the LayerNorm sums at the end of: the separate kernels
            d_y1, df = ops.add_layernorm_bwd(dy, t["z2"], t["mean2"], t["rstd2"], layer.norm2.gamma, p, rng,
                                             layer.site_res2, layer.norm2.gamma.grad, layer.norm2.beta.grad)
            ffn_bwd(df, t["y1"], M, layer, (t["hpre"], t["h"]), p, rng, d_y1)
            dx, da = ops.add_layernorm_bwd(d_y1, t["z1"], t["mean1"], t["rstd1"], layer.norm1.gamma, p, rng,
                                           layer.site_res1, layer.norm1.gamma.grad, layer.norm1.beta.grad)
            mha_block_bwd(da, x2, x2, attn, s_att, B, T, T, key_mask, attn.band, dx, True, dx, True)
            return dx
        df, dh, da, dx, dctx = r
        lin_bwd_w(t["h"], df, M, layer.ffn1, dff, d)
        lin_bwd_w(t["y1"], dh, M, layer.ffn0, d, dff)
        lin_bwd_w(s_att[3], da, M, attn.dense, d, d)
        mha_block_bwd(None, x2, x2, attn, s_att, B, T, T, key_mask, attn.band, dx, True, dx, True, dctx=dctx,
                      hand_over=(lambda dqkv: hand_over(dx, dqkv)) if hand_over is not None else None)
    return dx


class EncoderLayerFn(torch.autograd.Function):
    """TransformerEncoderLayer.call (reference athena/layers/transformer.py:296-310)."""

    @staticmethod
    def forward(ctx, x, layer, key_mask, training, *params):
        B, T, d = x.shape
        M = B * T
        rng = layer.store.rng_state
        p = layer.rate if training else 0.0
        x2 = x.reshape(M, d)
        ctx.layer, ctx.key_mask, ctx.p, ctx.shape = layer, key_mask, p, (B, T, d)
        ctx.tail = None
        if ops.layer_tail_ok(x2, d, layer.self_attn.dense, layer.ffn0, layer.ffn1):
            # everything after the attention kernel in one launch -- and, when the encoder stack has
            # linked the layers, the q / k / v projections of the layer behind this one as well, which
            # that layer then finds waiting (its backward pass is the same either way: it keeps q, k, v)
            x2 = x2.contiguous()
            pre = layer.__dict__.pop("_pre_qkv", None)
            if pre is not None and (pre[0] != x2.data_ptr() or pre[1] != (M, d)):
                pre = None
            c, s_att = mha_block_fwd(x2, x2, layer.self_attn, B, T, T, key_mask, layer.self_attn.band, project=False,
                                     qkv=pre[2] if pre is not None else None)
            nxt = layer.__dict__.get("_next_in_stack")
            nxt = nxt[0] if nxt else None
            nq = None
            if nxt is not None and ops.layer_tail_ok(x2, d, nxt.self_attn.dense, nxt.ffn0, nxt.ffn1) and \
                    all(m.tiled_t is not None for m in (nxt.self_attn.wq, nxt.self_attn.wk, nxt.self_attn.wv)) and \
                    os.environ.get("ATHENA_TAIL_QKV", "1") != "0":
                nq = (nxt.self_attn.wq, nxt.self_attn.wk, nxt.self_attn.wv)
            # the residual stream passes from tail to tail in fp32 (the layer in front left a copy)
            x32 = layer.__dict__.pop("_pre_x32", None)
            if x32 is not None and (x32[0] != x2.data_ptr() or x32[1] != (M, d)):
                x32 = None
            keep32 = nxt is not None and os.environ.get("ATHENA_TAIL_FP32_STREAM", "1") != "0" and \
                ops.layer_tail_ok(x2, d, nxt.self_attn.dense, nxt.ffn0, nxt.ffn1)
            t = ops.layer_tail_fwd(c, x2, layer.self_attn.dense, layer.norm1, layer.ffn0, layer.ffn1,
                                   layer.norm2, LN_EPS, p, rng, layer.site_res1, layer.site_ffn_h, layer.site_res2,
                                   next_qkv=nq, x32=x32[2] if x32 is not None else None, want_y32=keep32)
            y2 = t.pop("y2")
            if keep32:
                nxt.__dict__["_pre_x32"] = (y2.data_ptr(), (M, d), t.pop("y32"))
            if nq is not None:
                nxt.__dict__["_pre_qkv"] = (y2.data_ptr(), (M, d), (t.pop("q"), t.pop("k"), t.pop("v")))
            ctx.tail = t
            ctx.saved = (x2, s_att)
            return y2.view(B, T, d)
        a, s_att = mha_block_fwd(x2, x2, layer.self_attn, B, T, T, key_mask, layer.self_attn.band)
        # (the residual stream passes from LayerNorm to LayerNorm in fp32 here too, where the kernels cover
        # the width)
        wide = ops.layernorm_wide_ok(a) and x2.is_contiguous() and \
            os.environ.get("ATHENA_TAIL_FP32_STREAM", "1") != "0"
        x32 = layer.__dict__.pop("_pre_x32", None)
        if x32 is not None and (not wide or x32[0] != x2.data_ptr() or x32[1] != (M, d)):
            x32 = None
        r1 = ops.add_layernorm_fwd(x2, a, layer.norm1.gamma, layer.norm1.beta, LN_EPS, p, rng, layer.site_res1,
                                   a32=x32[2] if x32 is not None else None, want_y32=wide)
        z1, y1, mean1, rstd1 = r1[:4]
        f, s_ffn = ffn_fwd(y1, M, layer, p, rng)
        nxt = layer.__dict__.get("_next_in_stack")
        keep32 = wide and bool(nxt)
        r2 = ops.add_layernorm_fwd(y1, f, layer.norm2.gamma, layer.norm2.beta, LN_EPS, p, rng, layer.site_res2,
                                   a32=r1[4] if wide else None, want_y32=keep32)
        z2, y2, mean2, rstd2 = r2[:4]
        if keep32:
            nxt[0].__dict__["_pre_x32"] = (y2.data_ptr(), (M, d), r2[4])
        ctx.saved = (x2, s_att, z1, y1, mean1, rstd1, s_ffn, z2, mean2, rstd2)
        return y2.view(B, T, d)

    @staticmethod
    def backward(ctx, dy):
        layer, key_mask, p = ctx.layer, ctx.key_mask, ctx.p
        B, T, d = ctx.shape
        M = B * T
        rng = layer.store.rng_state
        dy = dy.contiguous().view(M, d)
        # The layer behind this one may have handed over the last product of its backward pass,
        # dq | dk | dv times its fused projection kernels, which is part of THIS layer's output
        # gradient: the one-launch tail adds it at its start; any other path adds it here and now.
        pend = layer.__dict__.pop("_pending_dx", None)
        if pend is not None and (ctx.tail is None or pend[0] != dy.data_ptr() or pend[1] != (M, d)):
            _add_pending(pend, dy)
            pend = None
        if ctx.tail is not None:
            t, ctx.tail = ctx.tail, None
            x2, s_att = ctx.saved
            ctx.saved = None
            # and this layer hands its own to the layer in front, if that one has a tail to add it in
            prev = layer.__dict__.get("_prev_in_stack")
            prev = prev[0] if prev else None
            hand_over = None
            if prev is not None and ctx.needs_input_grad[0] and os.environ.get("ATHENA_TAIL_QKV", "1") != "0" and \
                    ops.layer_tail_ok(x2, d, prev.self_attn.dense, prev.ffn0, prev.ffn1) and \
                    all(m.tiled is not None for m in (layer.self_attn.wq, layer.self_attn.wk, layer.self_attn.wv)):
                def hand_over(dx, dqkv, prev=prev, attn=layer.self_attn):
                    prev.__dict__["_pending_dx"] = (dx.data_ptr(), tuple(dx.shape), dqkv, attn)
            dx = _encoder_tail_bwd(dy, t, x2, s_att, layer, B, T, key_mask, p, rng, pend, hand_over)
            return (dx.view(B, T, d), None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)
        x2, s_att, z1, y1, mean1, rstd1, s_ffn, z2, mean2, rstd2 = ctx.saved
        ctx.saved = None
        with weight_grad_group():
            d_y1, df = ops.add_layernorm_bwd(dy, z2, mean2, rstd2, layer.norm2.gamma, p, rng, layer.site_res2,
                                             layer.norm2.gamma.grad, layer.norm2.beta.grad)
            ffn_bwd(df, y1, M, layer, s_ffn, p, rng, d_y1)
            dx, da = ops.add_layernorm_bwd(d_y1, z1, mean1, rstd1, layer.norm1.gamma, p, rng, layer.site_res1,
                                           layer.norm1.gamma.grad, layer.norm1.beta.grad)
            mha_block_bwd(da, x2, x2, layer.self_attn, s_att, B, T, T, key_mask, layer.self_attn.band, dx, True, dx, True)
        return (dx.view(B, T, d), None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class _MemoryGrad:
    """Gradient of the encoder output collected over the decoder layers that attend to it: the
    first layer to run its backward pass stores into a shared buffer, the others add to it inside
    their kernels, and the last one hands the buffer to autograd (the others return None), so no
    separate additions are launched.  The bookkeeping hangs on the encoder-output tensor object
    the layers of one forward pass share, and goes away with it."""

    @staticmethod
    def enter(memory):
        slot = getattr(memory, "_athena_grad_slot", None)
        if slot is None:
            slot = {"users": 0, "buf": None}
            memory._athena_grad_slot = slot
        slot["users"] += 1
        return slot

    @staticmethod
    def leave(slot):
        """True if the caller is the last user (it returns the buffer)."""
        slot["users"] -= 1
        return slot["users"] == 0


class DecoderLayerFn(torch.autograd.Function):
    """TransformerDecoderLayer.call (reference athena/layers/transformer.py:374-388)."""

    @staticmethod
    def forward(ctx, tgt, memory, layer, tgt_key_mask, mem_key_mask, training, *params):
        B, L, d = tgt.shape
        T = memory.shape[1]
        M = B * L
        rng = layer.store.rng_state
        p = layer.rate if training else 0.0
        t2 = tgt.reshape(M, d)
        m2 = memory.reshape(B * T, d)
        ctx.layer, ctx.masks, ctx.p, ctx.shape = layer, (tgt_key_mask, mem_key_mask), p, (B, L, T, d)
        ctx.mem_slot = _MemoryGrad.enter(memory) if ctx.needs_input_grad[1] else None
        ctx.tails = None
        if os.environ.get("ATHENA_LAYER_TAIL_DECODER", "1") != "0" and \
                ops.layer_tail_ok(t2, d, layer.attn1.dense) and \
                ops.layer_tail_ok(t2, d, layer.attn2.dense, layer.ffn0, layer.ffn1):
            # two launches besides the attention kernels and their q/k/v projections: self-attention
            # projection + LayerNorm, then cross-attention projection + LayerNorm + FFN + LayerNorm
            t2 = t2.contiguous()
            # (the self-attention projections may be waiting: the second tail of the layer in front ran them)
            pre = layer.__dict__.pop("_pre_qkv", None)
            if pre is not None and (pre[0] != t2.data_ptr() or pre[1] != (M, d)):
                pre = None
            c1, s1 = mha_block_fwd(t2, t2, layer.attn1, B, L, L, tgt_key_mask, True, project=False,
                                   qkv=pre[2] if pre is not None else None)
            # the first tail also projects its output for the cross-attention's queries; the keys and
            # values of the memory may be waiting (the decoder stack projects them for all layers at once)
            fold = os.environ.get("ATHENA_TAIL_QKV", "1") != "0" and layer.attn2.wq.tiled_t is not None
            # (the residual stream passes from tail to tail in fp32, as in the encoder)
            f32 = os.environ.get("ATHENA_TAIL_FP32_STREAM", "1") != "0"
            x32 = layer.__dict__.pop("_pre_x32", None)
            if x32 is not None and (x32[0] != t2.data_ptr() or x32[1] != (M, d)):
                x32 = None
            ta = ops.layer_tail_fwd(c1, t2, layer.attn1.dense, layer.norm1, None, None, None, LN_EPS, p, rng,
                                    layer.site_res1, 0, 0, next_qkv=(layer.attn2.wq,) if fold else None,
                                    x32=x32[2] if x32 is not None else None, want_y32=f32)
            y1 = ta["y1"]
            y1_32 = ta.pop("y32", None)
            q2 = ta.pop("q", None)
            kv = layer.__dict__.pop("_pre_kv", None)
            if kv is not None and (kv[0] != m2.data_ptr() or kv[1] != (B * T, d)):
                kv = None
            c2, s2 = mha_block_fwd(y1, m2, layer.attn2, B, L, T, mem_key_mask, False, project=False,
                                   qkv=(q2, kv[2][0] if kv else None, kv[2][1] if kv else None))
            nxt = layer.__dict__.get("_next_in_stack")
            nxt = nxt[0] if nxt else None
            nq = None
            if nxt is not None and ops.layer_tail_ok(t2, d, nxt.attn1.dense) and \
                    all(m.tiled_t is not None for m in (nxt.attn1.wq, nxt.attn1.wk, nxt.attn1.wv)) and \
                    os.environ.get("ATHENA_TAIL_QKV", "1") != "0":
                nq = (nxt.attn1.wq, nxt.attn1.wk, nxt.attn1.wv)
            keep32 = f32 and nxt is not None and ops.layer_tail_ok(t2, d, nxt.attn1.dense)
            tb = ops.layer_tail_fwd(c2, y1, layer.attn2.dense, layer.norm2, layer.ffn0, layer.ffn1, layer.norm3,
                                    LN_EPS, p, rng, layer.site_res2, layer.site_ffn_h, layer.site_res3, next_qkv=nq,
                                    x32=y1_32, want_y32=keep32)
            y3 = tb.pop("y2")
            if keep32:
                nxt.__dict__["_pre_x32"] = (y3.data_ptr(), (M, d), tb.pop("y32"))
            if nq is not None:
                nxt.__dict__["_pre_qkv"] = (y3.data_ptr(), (M, d), (tb.pop("q"), tb.pop("k"), tb.pop("v")))
            ctx.tails = (ta, tb)
            ctx.saved = (t2, m2, s1, s2)
            if ctx.mem_slot is not None:
                ctx.mem_slot["kv_fwd"] = ctx.mem_slot.get("kv_fwd", 0) + 1
            return y3.view(B, L, d)
        a1, s1 = mha_block_fwd(t2, t2, layer.attn1, B, L, L, tgt_key_mask, True)
        wide = ops.layernorm_wide_ok(a1) and t2.is_contiguous() and \
            os.environ.get("ATHENA_TAIL_FP32_STREAM", "1") != "0"
        x32 = layer.__dict__.pop("_pre_x32", None)
        if x32 is not None and (not wide or x32[0] != t2.data_ptr() or x32[1] != (M, d)):
            x32 = None
        r1 = ops.add_layernorm_fwd(t2, a1, layer.norm1.gamma, layer.norm1.beta, LN_EPS, p, rng, layer.site_res1,
                                   a32=x32[2] if x32 is not None else None, want_y32=wide)
        z1, y1, mean1, rstd1 = r1[:4]
        a2, s2 = mha_block_fwd(y1, m2, layer.attn2, B, L, T, mem_key_mask, False)
        r2 = ops.add_layernorm_fwd(y1, a2, layer.norm2.gamma, layer.norm2.beta, LN_EPS, p, rng, layer.site_res2,
                                   a32=r1[4] if wide else None, want_y32=wide)
        z2, y2, mean2, rstd2 = r2[:4]
        f, s_ffn = ffn_fwd(y2, M, layer, p, rng)
        nxt = layer.__dict__.get("_next_in_stack")
        keep32 = wide and bool(nxt)
        r3 = ops.add_layernorm_fwd(y2, f, layer.norm3.gamma, layer.norm3.beta, LN_EPS, p, rng, layer.site_res3,
                                   a32=r2[4] if wide else None, want_y32=keep32)
        z3, y3, mean3, rstd3 = r3[:4]
        if keep32:
            nxt[0].__dict__["_pre_x32"] = (y3.data_ptr(), (M, d), r3[4])
        ctx.saved = (t2, m2, s1, z1, y1, mean1, rstd1, s2, z2, y2, mean2, rstd2, s_ffn, z3, mean3, rstd3)
        return y3.view(B, L, d)

    @staticmethod
    def backward(ctx, dy):
        layer, p = ctx.layer, ctx.p
        tgt_key_mask, mem_key_mask = ctx.masks
        B, L, T, d = ctx.shape
        M = B * L
        rng = layer.store.rng_state
        dy = dy.contiguous().view(M, d)
        # (as in the encoder: the layer behind may have handed over its self-attention's last product)
        pend = layer.__dict__.pop("_pending_dx", None)
        if pend is not None and (ctx.tails is None or pend[0] != dy.data_ptr() or pend[1] != (M, d)):
            _add_pending(pend, dy)
            pend = None
        if ctx.tails is not None:
            (ta, tb), ctx.tails = ctx.tails, None
            t2, m2, s1, s2 = ctx.saved
            ctx.saved = None
            dff = layer.ffn0.out_features
            add = (pend[2], (pend[3].wq, pend[3].wk, pend[3].wv)) if pend is not None else None
            rb = ops.layer_tail_bwd(dy, tb, layer.attn2.dense, layer.norm2, layer.ffn0, layer.ffn1, layer.norm3, p, rng,
                                    layer.site_res2, layer.site_ffn_h, layer.site_res3, add=add)
            if rb is None:
                raise RuntimeError("decoder layer: the one-launch backward pass needs a running autograd pass")
            df, dh, da2, d_y1, dctx2 = rb
            y1 = ta["y1"]
            lin_bwd_w(tb["h"], df, M, layer.ffn1, dff, d)
            lin_bwd_w(tb["y1"], dh, M, layer.ffn0, d, dff)
            lin_bwd_w(s2[3], da2, M, layer.attn2.dense, d, d)
            slot = ctx.mem_slot
            # All layers' dk | dv side by side in one buffer, and ONE product with the stacked key / value
            # kernels when the last of them is in, instead of a product per layer added into the buffer
            # (six launches of 8000 x 256 x 512 on the decoder's chain of small launches became one of
            # 8000 x 256 x 3072).  Needs every decoder layer of the pass on this path.
            stack = layer.__dict__.get("_kv_stack")
            wide = layer.store.stack_view("decoder.memory_kv") if stack is not None and slot is not None and \
                slot.get("kv_fwd", 0) == stack[1] and m2.dtype == torch.bfloat16 and \
                os.environ.get("ATHENA_DEC_KV_STACK", "1") != "0" else None
            dkv_out = None
            if wide is not None:
                Kall = stack[1] * 2 * d
                if slot.get("dkv_all") is None:
                    slot["dkv_all"] = torch.empty((B * T, Kall), dtype=m2.dtype, device=m2.device)
                    slot["kv_left"] = stack[1]
                dkv_out = (slot["dkv_all"][:, stack[0] * 2 * d:(stack[0] + 1) * 2 * d], Kall)
                dmem, acc_mem = None, False
            elif slot is None:
                dmem, acc_mem = torch.empty((B * T, d), dtype=m2.dtype, device=m2.device), False
            else:
                acc_mem = slot["buf"] is not None
                if not acc_mem:
                    slot["buf"] = torch.empty((B * T, d), dtype=m2.dtype, device=m2.device)
                dmem = slot["buf"]
            # the cross-attention's dq Wq^T is part of the gradient of y1: the backward tail of the
            # self-attention sub-layer adds it at its start instead of a launch of its own
            held = []
            fold_q = os.environ.get("ATHENA_TAIL_QKV", "1") != "0" and layer.attn2.wq.tiled is not None
            mha_block_bwd(None, y1, m2, layer.attn2, s2, B, L, T, mem_key_mask, False, d_y1, True, dmem, acc_mem,
                          dctx=dctx2, hand_over=held.append if fold_q else None, dkv_out=dkv_out)
            if wide is not None:
                slot["kv_left"] -= 1
                if slot["kv_left"] == 0:
                    acc_mem = slot["buf"] is not None
                    if not acc_mem:
                        slot["buf"] = torch.empty((B * T, d), dtype=m2.dtype, device=m2.device)
                    peers = layer.__dict__.get("_kv_peers")
                    if ops.PANEL_ENABLED and d == 256 and peers is not None and len(peers) == stack[1] and \
                            2 * len(peers) <= ops.PANEL_MAX_BLOCKS and \
                            all(a.wk.tiled is not None and a.wv.tiled is not None for a in peers):
                        ops.panel_gemm_k(slot["dkv_all"], B * T, Kall, [(w.tiled, d) for a in peers for w in (a.wk, a.wv)],
                                         slot["buf"], d, accumulate=acc_mem)
                    else:
                        ops.gemm(slot["dkv_all"], wide, slot["buf"], B * T, d, Kall, trans_b=True, lda=Kall, ldb=Kall,
                                 ldc=d, accumulate=acc_mem)
                    slot["dkv_all"] = None
                dmem = slot["buf"]
            if slot is not None and not _MemoryGrad.leave(slot):
                dmem = None
            ra = ops.layer_tail_bwd(d_y1, ta, layer.attn1.dense, layer.norm1, None, None, None, p, rng,
                                    layer.site_res1, 0, 0, add=(held[0], (layer.attn2.wq,)) if held else None)
            if ra is None:
                raise RuntimeError("decoder layer: the one-launch backward pass needs a running autograd pass")
            _, _, da1, dtgt, dctx1 = ra
            lin_bwd_w(s1[3], da1, M, layer.attn1.dense, d, d)
            prev = layer.__dict__.get("_prev_in_stack")
            prev = prev[0] if prev else None
            hand_over = None
            if prev is not None and ctx.needs_input_grad[0] and os.environ.get("ATHENA_TAIL_QKV", "1") != "0" and \
                    os.environ.get("ATHENA_LAYER_TAIL_DECODER", "1") != "0" and \
                    ops.layer_tail_ok(t2, d, prev.attn2.dense, prev.ffn0, prev.ffn1) and \
                    all(m.tiled is not None for m in (layer.attn1.wq, layer.attn1.wk, layer.attn1.wv)):
                def hand_over(dqkv, prev=prev, attn=layer.attn1, dtgt=dtgt):
                    prev.__dict__["_pending_dx"] = (dtgt.data_ptr(), tuple(dtgt.shape), dqkv, attn)
            mha_block_bwd(None, t2, t2, layer.attn1, s1, B, L, L, tgt_key_mask, True, dtgt, True, dtgt, True, dctx=dctx1,
                          hand_over=hand_over)
            return (dtgt.view(B, L, d), None if dmem is None else dmem.view(B, T, d), None, None, None, None) + \
                (None,) * (len(ctx.needs_input_grad) - 6)
        t2, m2, s1, z1, y1, mean1, rstd1, s2, z2, y2, mean2, rstd2, s_ffn, z3, mean3, rstd3 = ctx.saved
        ctx.saved = None
        with weight_grad_group():
            d_y2, df = ops.add_layernorm_bwd(dy, z3, mean3, rstd3, layer.norm3.gamma, p, rng, layer.site_res3,
                                             layer.norm3.gamma.grad, layer.norm3.beta.grad)
            ffn_bwd(df, y2, M, layer, s_ffn, p, rng, d_y2)
            d_y1, da2 = ops.add_layernorm_bwd(d_y2, z2, mean2, rstd2, layer.norm2.gamma, p, rng, layer.site_res2,
                                              layer.norm2.gamma.grad, layer.norm2.beta.grad)
            slot = ctx.mem_slot
            if slot is None:
                dmem, acc_mem = torch.empty((B * T, d), dtype=m2.dtype, device=m2.device), False
            else:
                acc_mem = slot["buf"] is not None
                if not acc_mem:
                    slot["buf"] = torch.empty((B * T, d), dtype=m2.dtype, device=m2.device)
                dmem = slot["buf"]
            mha_block_bwd(da2, y1, m2, layer.attn2, s2, B, L, T, mem_key_mask, False, d_y1, True, dmem, acc_mem)
            if slot is not None and not _MemoryGrad.leave(slot):
                dmem = None
            dtgt, da1 = ops.add_layernorm_bwd(d_y1, z1, mean1, rstd1, layer.norm1.gamma, p, rng, layer.site_res1,
                                              layer.norm1.gamma.grad, layer.norm1.beta.grad)
            mha_block_bwd(da1, t2, t2, layer.attn1, s1, B, L, L, tgt_key_mask, True, dtgt, True, dtgt, True)
        return (dtgt.view(B, L, d), None if dmem is None else dmem.view(B, T, d), None, None, None, None) + \
            (None,) * (len(ctx.needs_input_grad) - 6)


class MultiHeadAttentionFn(torch.autograd.Function):
    """Stand-alone MultiHeadAttention.call(v, k, q, mask) with k is v (attention.py:147-171)."""

    @staticmethod
    def forward(ctx, q_in, kv_in, attn, key_mask, causal, *params):
        B, Tq, d = q_in.shape
        Tk = kv_in.shape[1]
        q2, kv2 = q_in.reshape(B * Tq, d), kv_in.reshape(B * Tk, d)
        out, saved = mha_block_fwd(q2, kv2, attn, B, Tq, Tk, key_mask, causal)
        ctx.attn, ctx.key_mask, ctx.causal, ctx.shape = attn, key_mask, causal, (B, Tq, Tk, d)
        ctx.same = q_in.data_ptr() == kv_in.data_ptr() and Tq == Tk
        ctx.saved = (q2, kv2, saved)
        return out.view(B, Tq, d)

    @staticmethod
    def backward(ctx, dout):
        B, Tq, Tk, d = ctx.shape
        q2, kv2, saved = ctx.saved
        ctx.saved = None
        dout = dout.contiguous().view(B * Tq, d)
        dq_in = torch.empty_like(q2)
        dkv_in = torch.empty_like(kv2)
        mha_block_bwd(dout, q2, kv2, ctx.attn, saved, B, Tq, Tk, ctx.key_mask, ctx.causal, dq_in, False, dkv_in,
                      False)
        return (dq_in.view(B, Tq, d), dkv_in.view(B, Tk, d), None, None, None) + \
            (None,) * (len(ctx.needs_input_grad) - 5)


class DenseFn(torch.autograd.Function):
    """y = x @ kernel + bias with an fp32 output whose rows are padded to a multiple of 8 columns
    (the vocabulary projections: speech_transformer.py:130,151 and mtl_seq2seq.py:62,83).
    Returns the [B, T, N] view of the padded buffer [B, T, Np].  The loss kernels hand back a
    gradient that is such a view too; the backward pass then uses its buffer in place instead of
    copying it into a fresh padded tensor."""

    @staticmethod
    def forward(ctx, x, dense, kernel, bias, joins_memory_grad=False):
        B, T, K = x.shape
        N = dense.out_features
        # ``x`` may be the encoder output that decoder layers of this forward pass attend to (they have
        # entered its gradient slot by now).  Inside a solver step, whose total loss reaches ``x`` through
        # them too, this projection's dx then goes into the buffer they share instead of into a tensor
        # of its own that autograd would add with one more launch.
        ctx.mem_slot = None
        if joins_memory_grad and ctx.needs_input_grad[0] and ops.root_gradient_is_one() and \
                getattr(x, "_athena_grad_slot", None) is not None:
            ctx.mem_slot = _MemoryGrad.enter(x)
        Np = ops.roundup(N, 8)
        x2 = x.reshape(B * T, K)
        # the padding columns [N, Np) stay unwritten: nothing reads them (the returned view ends at N,
        # the loss kernels take V columns of each row)
        out, _ = lin_fwd(x2, B * T, dense, out_dtype=torch.float32, ldc=Np)
        ctx.dense, ctx.x2, ctx.shape = dense, x2, (B, T, K, N, Np)
        return out.view(B, T, Np)[..., :N]

    @staticmethod
    def backward(ctx, dy):
        dense, x2 = ctx.dense, ctx.x2
        B, T, K, N, Np = ctx.shape
        M = B * T
        g16 = ops.take_bf16_grad(dy)
        if g16 is not None:
            # the loss wrote its gradient in bf16 into a padded buffer (padding columns zeroed)
            if g16.dtype != x2.dtype or g16.numel() != M * Np:
                raise RuntimeError("DenseFn: bf16 gradient buffer does not match the projection")
            dy = g16.view(M, Np)
        elif Np != N and dy.stride() == (T * Np, Np, 1) and dy.storage_offset() % 4 == 0 and \
                dy.untyped_storage().nbytes() >= (dy.storage_offset() + M * Np) * dy.element_size():
            # a view of a padded buffer (what the loss kernels return): take the buffer itself
            dy = dy.as_strided((M, Np), (Np, 1))
            dy[:, N:].zero_()
        elif Np != N:
            full = torch.zeros((M, Np), dtype=dy.dtype, device=dy.device)
            full[:, :N] = dy.reshape(M, N)
            dy = full
        else:
            dy = dy.contiguous().view(M, Np)
        if x2.dtype != dy.dtype:
            g = torch.empty((M, Np), dtype=x2.dtype, device=dy.device)
            ops.cast(dy, g)
            dy = g
        lin_bwd_w(x2, dy, M, dense, K, Np)
        slot, ctx.mem_slot = ctx.mem_slot, None
        ctx.x2 = None
        if slot is not None and slot["buf"] is None and dy.dtype == x2.dtype:
            # first to arrive: the shared buffer starts as this product
            slot["buf"] = dx = lin_bwd_x(dy, M, dense, Np)
            return (dx.view(B, T, K) if _MemoryGrad.leave(slot) else None), None, None, None, None
        dx = lin_bwd_x(dy, M, dense, Np)
        if slot is not None:
            if slot["buf"].dtype == dx.dtype:
                dx = slot["buf"].add_(dx)
                if not _MemoryGrad.leave(slot):
                    dx = None
            elif _MemoryGrad.leave(slot):
                dx = dx + slot["buf"].to(dx.dtype)
        return (None if dx is None else dx.view(B, T, K)), None, None, None, None


class XNetFn(torch.autograd.Function):
    """x_net: Conv2D-BN-relu6 x2, reshape, Dense+relu6, +PE, dropout
    (reference athena/models/speech_transformer.py:72-105)."""

    @staticmethod
    def forward(ctx, x0, net, training, *params):
        st = net.store
        cd = st.compute_dtype
        dev = x0.device
        B, T, Fd, nc = x0.shape
        C, d = net.num_filters, net.d_model
        T1, F1 = ops.same_out(T), ops.same_out(Fd)
        T2, F2 = ops.same_out(T1), ops.same_out(F1)
        rows1, rows2 = B * T1 * F1, B * T2 * F2
        p = net.rate if training else 0.0
        x0 = x0.contiguous()
        stats = torch.empty(2 * C, dtype=torch.float64, device=dev) if training else None
        mr1 = mr2 = mom1 = c1 = None
        a1 = torch.empty((rows1, C), dtype=cd, device=dev)
        fused1 = nc == 1 and C % 8 == 0 and 256 % (C // 8) == 0
        if fused1:
            # one input channel: conv + BN + relu6 in one pass, statistics in closed form from the
            # input moments; the pre-activation tensor is never stored
            if training:
                mom1 = torch.empty(ops.conv1_moments_len(), dtype=torch.float64, device=dev)
                mr1 = torch.empty(2 * C, dtype=torch.float32, device=dev)
                ops.conv1_bn_stats(x0, net.conv1.kernel, B, T, Fd, C, mom1, stats)
                ops.bn_finalize(stats, rows1, C, BN_EPS, BN_MOMENTUM, mr1, net.bn1.moving_mean,
                                net.bn1.moving_variance)
                ops.conv1_bn_relu6_fwd(x0, net.conv1.kernel, mr1, None, net.bn1.gamma, net.bn1.beta, BN_EPS, False, B,
                                       T, Fd, C, a1)
            else:
                ops.conv1_bn_relu6_fwd(x0, net.conv1.kernel, net.bn1.moving_mean, net.bn1.moving_variance,
                                       net.bn1.gamma, net.bn1.beta, BN_EPS, True, B, T, Fd, C, a1)
        else:
            # a few input channels: direct kernel on the fp32 input, then BN + relu6
            c1 = torch.empty((rows1, C), dtype=cd, device=dev)
            ops.conv_small_cin_fwd(x0, net.conv1.kernel, c1, B, T, Fd, nc, C, stats)
            if training:
                mr1 = torch.empty(2 * C, dtype=torch.float32, device=dev)
                ops.bn_finalize(stats, rows1, C, BN_EPS, BN_MOMENTUM, mr1, net.bn1.moving_mean,
                                net.bn1.moving_variance)
                ops.bn_relu6_apply(c1, mr1, net.bn1.gamma, net.bn1.beta, rows1, C, a1)
            else:
                ops.bn_relu6_infer(c1, net.bn1.moving_mean, net.bn1.moving_variance, net.bn1.gamma, net.bn1.beta,
                                   BN_EPS, rows1, C, a1)
        # conv2 (implicit GEMM)
        c2 = torch.empty((rows2, C), dtype=cd, device=dev)
        if cd == torch.bfloat16:
            ops.conv3x3s2_fwd(a1, net.conv2.shadow_t, 1, c2, B, T1, F1, C, C, stats)
        else:
            ops.conv3x3s2_fwd(a1, net.conv2.kernel, 0, c2, B, T1, F1, C, C, stats)
        a2 = torch.empty_like(c2)
        if training:
            mr2 = torch.empty(2 * C, dtype=torch.float32, device=dev)
            ops.bn_finalize(stats, rows2, C, BN_EPS, BN_MOMENTUM, mr2, net.bn2.moving_mean, net.bn2.moving_variance)
            ops.bn_relu6_apply(c2, mr2, net.bn2.gamma, net.bn2.beta, rows2, C, a2)
        else:
            ops.bn_relu6_infer(c2, net.bn2.moving_mean, net.bn2.moving_variance, net.bn2.gamma, net.bn2.beta,
                               BN_EPS, rows2, C, a2)
        # Reshape(-1, F2*C) is a view of the NHWC tensor; Dense + relu6; + PE; dropout
        M = B * T2
        y, _ = lin_fwd(a2.view(M, F2 * C), M, net.dense, act=ACT_RELU6)
        out = torch.empty_like(y)
        ops.add_pe_dropout(y, st.pe(d), B, T2, d, 1.0, p, st.rng_state, net.site_drop, out)
        ctx.net, ctx.p, ctx.dims = net, p, (B, T, Fd, nc, C, d, T1, F1, T2, F2)
        ctx.saved = (x0, c1, a1, mr1, mom1, c2, a2, mr2, y) if training else None
        return out.view(B, T2, d)

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise RuntimeError("x_net: backward needs a training-mode forward")
        net, p = ctx.net, ctx.p
        st = net.store
        B, T, Fd, nc, C, d, T1, F1, T2, F2 = ctx.dims
        x0, c1, a1, mr1, mom1, c2, a2, mr2, y = ctx.saved
        ctx.saved = None
        dev = dout.device
        rows1, rows2 = B * T1 * F1, B * T2 * F2
        M = B * T2
        dout = dout.contiguous().view(M, d)
        dpre = torch.empty_like(dout)
        ops.dropout_bwd(dout, 1.0, p, st.rng_state, net.site_drop, y, dpre)
        a2m = a2.view(M, F2 * C)
        lin_bwd_w(a2m, dpre, M, net.dense, F2 * C, d)
        da2 = lin_bwd_x(dpre, M, net.dense, d)
        scratch = ops.bn_relu6_bwd_scratch(max(rows1, rows2), C, dev)
        dc2 = torch.empty_like(c2)
        ops.bn_relu6_bwd(da2, c2, mr2, net.bn2.gamma, net.bn2.beta, rows2, C, scratch, dc2, net.bn2.gamma.grad,
                         net.bn2.beta.grad)
        ops.conv3x3s2_wgrad(a1, dc2, net.conv2.kernel.grad, B, T1, F1, C, C)
        da1 = torch.empty_like(a1)
        w_hwio = net.conv2.shadow if c2.dtype == torch.bfloat16 else net.conv2.kernel
        if mom1 is not None:
            # the weight, gamma and beta gradients of the fused first layer need only tap sums of the
            # gated gradient; the relu6 gate is read off a1 while da1 is streamed through
            ops.conv3x3s2_dgrad(dc2, w_hwio, da1, B, T1, F1, C, C)
            ops.conv1_bn_bwd(x0, da1, net.conv1.kernel, mom1, mr1, net.bn1.gamma, B, T, Fd, C, net.conv1.kernel.grad,
                             net.bn1.gamma.grad, net.bn1.beta.grad, relu6_out=a1, beta=net.bn1.beta)
        else:
            ops.conv3x3s2_dgrad(dc2, w_hwio, da1, B, T1, F1, C, C)
            dc1 = torch.empty_like(c1)
            ops.bn_relu6_bwd(da1, c1, mr1, net.bn1.gamma, net.bn1.beta, rows1, C, scratch, dc1, net.bn1.gamma.grad,
                             net.bn1.beta.grad)
            ops.conv_small_cin_wgrad(x0, dc1, net.conv1.kernel.grad, B, T, Fd, nc, C)
        return (None, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class YNetFn(torch.autograd.Function):
    """y_net: Embedding, *sqrt(d) + PE, dropout (speech_transformer.py:108-113)."""

    @staticmethod
    def forward(ctx, ids, net, training, table):
        st = net.store
        B, L = ids.shape
        d, V = net.d_model, net.num_class
        p = net.rate if training else 0.0
        ids = ids.contiguous()
        out = torch.empty((B, L, d), dtype=st.compute_dtype, device=ids.device)
        ops.embed_fwd(table, ids, st.pe(d), B, L, d, V, net.xscale, p, st.rng_state, net.site_drop, out)
        ctx.net, ctx.ids, ctx.p = net, ids, p
        return out

    @staticmethod
    def backward(ctx, dout):
        net, ids, p = ctx.net, ctx.ids, ctx.p
        st = net.store
        B, L = ids.shape
        g = net.embedding.grad          # every row is written
        ops.embed_bwd(dout.contiguous(), ids, B, L, net.d_model, net.num_class, net.xscale, p, st.rng_state,
                      net.site_drop, g)
        return None, None, None, None
