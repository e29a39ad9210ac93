"""Building blocks of the DeepSpeech model (reference athena/models/deep_speech.py:45-82): the two
wide-kernel convolutions with BatchNormalization + relu6, the recurrent layers (GRUCell / LSTMCell
under tf.keras.layers.RNN, athena/layers/commons.py:486-491) each followed by BatchNormalization, and
a Dense layer with an activation.  Every block is one autograd Function over the C ABI."""
import torch
from torch import nn

from .. import ops
from ..ops import ACT_MUL_DRELU6, ACT_NONE, ACT_RELU6
from ..runtime import constant, glorot_uniform
from .commons import BatchNormParams, Dense, _Bound
from .fused import BN_EPS, BN_MOMENTUM, _WgradQueue, lin_bwd_w, lin_bwd_x, lin_fwd

SUPPORTED_RNNS = {"lstm": ops.RNN_LSTM, "gru": ops.RNN_GRU, "cudnnlstm": ops.RNN_LSTM, "cudnngru": ops.RNN_GRU}


def orthogonal():
    """tf.keras.initializers.Orthogonal: Q of the QR decomposition of a normal matrix (sign-fixed)."""
    def f(shape, gen):
        rows, cols = shape[0], int(torch.tensor(shape[1:]).prod())
        a = torch.randn((max(rows, cols), min(rows, cols)), generator=gen, dtype=torch.float64)
        q, r = torch.linalg.qr(a)
        q = q * torch.sign(torch.diagonal(r))
        if rows < cols:
            q = q.t()
        return q.reshape(shape)
    return f


def lstm_bias():
    """Keras LSTMCell bias: zeros with ones on the forget gate (unit_forget_bias=True)."""
    def f(shape, gen):
        h = shape[0] // 4
        b = torch.zeros(shape, dtype=torch.float64)
        b[h:2 * h] = 1.0
        return b
    return f


class Conv2DParams(_Bound):
    """Kernel of a Conv2D(kernel_size=(KH, KW), no bias): HWIO [KH, KW, Cin, Cout], glorot-uniform."""

    def __init__(self, store, name, kh, kw, cin, cout):
        super().__init__()
        self.store, self.vname = store, name
        self.kh, self.kw, self.cin, self.cout = kh, kw, cin, cout
        store.register(name + ".kernel", (kh, kw, cin, cout), glorot_uniform(kh * kw * cin, kh * kw * cout),
                       matrix=(kh * kw * cin, cout))
        self.shadow = self.shadow_t = None

    def bind(self):
        self._bind_param("kernel", self.vname + ".kernel")
        if self.store.shadow is not None:
            self.shadow, self.shadow_t = self.store.shadow_views(self.vname + ".kernel")

    def hwio(self, dtype):
        """The kernel in the compute dtype, HWIO."""
        return self.shadow if dtype == torch.bfloat16 else self.kernel


class RNNParams(_Bound):
    """Variables of a GRUCell / LSTMCell in Keras' shapes: kernel [in, G*H] (glorot-uniform),
    recurrent_kernel [H, G*H] (orthogonal), bias [2, 3H] (GRU, reset_after: input and recurrent
    bias) or [4H] (LSTM, forget gate at one)."""

    def __init__(self, store, name, cell, in_features, hidden):
        super().__init__()
        self.store, self.vname, self.cell = store, name, cell
        self.in_features, self.hidden = in_features, hidden
        self.gates = 4 if cell == ops.RNN_LSTM else 3
        g = self.gates * hidden
        store.register(name + ".kernel", (in_features, g), glorot_uniform(in_features, g), matrix=(in_features, g))
        store.register(name + ".recurrent_kernel", (hidden, g), orthogonal(), matrix=(hidden, g))
        if cell == ops.RNN_LSTM:
            store.register(name + ".bias", (g,), lstm_bias())
        else:
            store.register(name + ".bias", (2, g), constant(0.0))
        self.k_shadow = self.k_shadow_t = self.u_shadow = self.u_shadow_t = None

    def bind(self):
        self._bind_param("kernel", self.vname + ".kernel")
        self._bind_param("recurrent_kernel", self.vname + ".recurrent_kernel")
        self._bind_param("bias", self.vname + ".bias")
        if self.store.shadow is not None:
            self.k_shadow, self.k_shadow_t = self.store.shadow_views(self.vname + ".kernel")
            self.u_shadow, self.u_shadow_t = self.store.shadow_views(self.vname + ".recurrent_kernel")

    def bias_views(self):
        """(input bias, recurrent bias or None, their gradient views)."""
        b, g = self.bias.detach(), self.bias.grad
        if self.cell == ops.RNN_LSTM:
            return b, None, g, None
        return b[0], b[1], g[0], g[1]


def _wgrad(x, dy, rows, K, N, ldx, lddy, kgrad, bgrad, table=True):
    """kgrad[K, N] = x[rows, K]^T dy[rows, N]; bgrad[N] = column sums of dy (when given).  bf16 operands
    inside a backward pass join the end-of-backward table launch (one workgroup per 256x256 output tile
    over the whole contraction: right for wide outputs, not for a small output over very many rows --
    ``table=False`` then takes the split-K product); everything else runs now."""
    if table and x.dtype == torch.bfloat16 and _WgradQueue.enabled and ldx % 8 == 0 and lddy % 8 == 0 and \
            ldx >= ops.roundup(K, 8) and lddy >= ops.roundup(N, 8) and x.data_ptr() % 16 == 0 and \
            dy.data_ptr() % 16 == 0 and _WgradQueue.push((x, dy, rows, K, N, ldx, lddy, kgrad, bgrad)):
        return
    ops.gemm(x, dy, kgrad, K, N, rows, trans_a=True, trans_b=False, lda=ldx, ldb=lddy, ldc=N, split_k=0,
             b_colsum=bgrad)


def _bn_train(x, rows, C, bn, stats, relu6):
    """Batch statistics (already in ``stats``), moving-average update, normalised output."""
    mr = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    ops.bn_finalize(stats, rows, C, BN_EPS, BN_MOMENTUM, mr, bn.moving_mean, bn.moving_variance)
    y = torch.empty_like(x)
    (ops.bn_relu6_apply if relu6 else ops.bn_apply)(x, mr, bn.gamma, bn.beta, rows, C, y)
    return y, mr


def _bn_infer(x, rows, C, bn, relu6):
    y = torch.empty_like(x)
    (ops.bn_relu6_infer if relu6 else ops.bn_infer)(x, bn.moving_mean, bn.moving_variance, bn.gamma, bn.beta, BN_EPS,
                                                    rows, C, y)
    return y


class ConvFrontFn(torch.autograd.Function):
    """Conv2D(41x11, stride 2x2)-BN-relu6, Conv2D(21x11, stride 2x1)-BN-relu6, Reshape to
    [B, T', F' * C] (deep_speech.py:49-71).  The first layer has a single input channel: its im2col
    matrix is built explicitly and multiplied as a Dense product; the second runs as an implicit GEMM."""

    @staticmethod
    def forward(ctx, x0, net, training, *params):
        st = net.store
        cd = st.compute_dtype
        dev = x0.device
        B, T, Fd, nc = x0.shape
        C = net.filters
        c1p, c2p = net.conv1, net.conv2
        T1, F1 = ops.conv2d_out_shape(T, Fd, c1p.kh, c1p.kw, 2, 2)
        T2, F2 = ops.conv2d_out_shape(T1, F1, c2p.kh, c2p.kw, 2, 1)
        rows1, rows2 = B * T1 * F1, B * T2 * F2
        x0 = x0.contiguous()
        K1 = c1p.kh * c1p.kw * nc
        ld1 = ops.roundup(K1, 8)
        col = torch.empty((rows1, ld1), dtype=cd, device=dev)
        ops.im2col(x0, col, ld1, B, T, Fd, nc, c1p.kh, c1p.kw, 2, 2)
        c1 = torch.empty((rows1, C), dtype=cd, device=dev)
        if cd == torch.bfloat16:
            ops.gemm(col, c1p.shadow_t, c1, rows1, C, ld1, trans_b=True, lda=ld1, ldb=c1p.shadow_t.shape[1], ldc=C)
        else:
            ops.gemm(col, c1p.kernel, c1, rows1, C, K1, trans_b=False, lda=ld1, ldb=C, ldc=C)
        stats = torch.empty(2 * C, dtype=torch.float64, device=dev) if training else None
        mr1 = mr2 = None
        if training:
            ops.bn_colstats(c1, rows1, C, stats)
            a1, mr1 = _bn_train(c1, rows1, C, net.bn1, stats, True)
        else:
            a1 = _bn_infer(c1, rows1, C, net.bn1, True)
        c2 = torch.empty((rows2, C), dtype=cd, device=dev)
        ops.conv2d_fwd(a1, c2p.hwio(cd), c2, B, T1, F1, C, C, c2p.kh, c2p.kw, 2, 1, stats)
        if training:
            a2, mr2 = _bn_train(c2, rows2, C, net.bn2, stats, True)
        else:
            a2 = _bn_infer(c2, rows2, C, net.bn2, True)
        ctx.net, ctx.dims = net, (B, T, Fd, nc, C, T1, F1, T2, F2, K1, ld1)
        ctx.saved = (col, c1, a1, mr1, c2, mr2) if training else None
        return a2.view(B, T2, F2 * C)

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise RuntimeError("deep speech front end: backward needs a training-mode forward")
        net = ctx.net
        B, T, Fd, nc, C, T1, F1, T2, F2, K1, ld1 = ctx.dims
        col, c1, a1, mr1, c2, mr2 = ctx.saved
        ctx.saved = None
        dev = dout.device
        rows1, rows2 = B * T1 * F1, B * T2 * F2
        c1p, c2p = net.conv1, net.conv2
        da2 = dout.contiguous().view(rows2, C)
        scratch = ops.bn_relu6_bwd_scratch(max(rows1, rows2), C, dev)
        dc2 = torch.empty_like(c2)
        ops.bn_relu6_bwd(da2, c2, mr2, net.bn2.gamma, net.bn2.beta, rows2, C, scratch, dc2, net.bn2.gamma.grad,
                         net.bn2.beta.grad)
        ops.conv2d_wgrad(a1, dc2, c2p.kernel.grad, B, T1, F1, C, C, c2p.kh, c2p.kw, 2, 1)
        da1 = torch.empty_like(a1)
        ops.conv2d_dgrad(dc2, c2p.hwio(c2.dtype), da1, B, T1, F1, C, C, c2p.kh, c2p.kw, 2, 1)
        dc1 = torch.empty_like(c1)
        ops.bn_relu6_bwd(da1, c1, mr1, net.bn1.gamma, net.bn1.beta, rows1, C, scratch, dc1, net.bn1.gamma.grad,
                         net.bn1.beta.grad)
        _wgrad(col, dc1, rows1, K1, C, ld1, C, c1p.kernel.grad.view(K1, C), None, table=False)
        return (None, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class ConvFront(nn.Module):
    def __init__(self, store, in_shape, filters):
        super().__init__()
        self.store, self.filters = store, filters
        feat, nc = int(in_shape[-2]), int(in_shape[-1])
        if filters % 8:
            raise ValueError("conv_filters must be a multiple of 8 (got %d)" % filters)
        self.feat_dim, self.in_channels = feat, nc
        self.conv1 = Conv2DParams(store, "conv2d", 41, 11, nc, filters)
        self.bn1 = BatchNormParams(store, "batch_normalization", filters)
        self.conv2 = Conv2DParams(store, "conv2d_1", 21, 11, filters, filters)
        self.bn2 = BatchNormParams(store, "batch_normalization_1", filters)
        self.out_dim = -(-feat // 2) * filters

    def forward(self, x0, training=None):
        if x0.shape[-2] != self.feat_dim or x0.shape[-1] != self.in_channels:
            raise ValueError("deep speech front end: input feature shape %s, expected (%d, %d)" %
                             (tuple(x0.shape[-2:]), self.feat_dim, self.in_channels))
        return ConvFrontFn.apply(x0, self, bool(training), *self.parameters())


class RnnBnFn(torch.autograd.Function):
    """tf.keras.layers.RNN(cell, return_sequences=True) + BatchNormalization (deep_speech.py:75-80).
    The input projection of all steps is one GEMM, the recurrence runs step by step inside the library,
    and the three weight gradients are GEMMs over all steps again."""

    @staticmethod
    def forward(ctx, x, layer, training, *params):
        p, bn = layer.rnn, layer.bn
        B, T, K = x.shape
        H, G, cell = p.hidden, p.gates, p.cell
        M = B * T
        dev, cd = x.device, x.dtype
        x2 = x.contiguous().view(M, K)
        b_in, b_rec, _, _ = p.bias_views()
        xw = torch.empty((M, G * H), dtype=cd, device=dev)
        if cd == torch.bfloat16:
            ops.gemm(x2, p.k_shadow_t, xw, M, G * H, K, trans_b=True, lda=K, ldb=p.k_shadow_t.shape[1], ldc=G * H,
                     bias=b_in)
            u = p.u_shadow
        else:
            ops.gemm(x2, p.kernel, xw, M, G * H, K, trans_b=False, lda=K, ldb=G * H, ldc=G * H, bias=b_in)
            u = p.recurrent_kernel
        hs_ext = torch.empty((B, T + 1, H), dtype=cd, device=dev)
        cs_ext = torch.empty((B, T + 1, H), dtype=torch.float32, device=dev) if cell == ops.RNN_LSTM else None
        rh = torch.empty((B, T, H), dtype=cd, device=dev) if cell == ops.RNN_GRU else None
        y = torch.empty((B, T, H), dtype=cd, device=dev)
        gates = torch.empty((B, T, G * H), dtype=cd, device=dev)
        ws = ops.rnn_workspace(cell, B, H, dev)
        ops.rnn_fwd(cell, xw, u, b_rec, hs_ext, cs_ext, y, gates, rh, B, T, H, ws,
                    u_t=p.u_shadow_t if cd == torch.bfloat16 else None)
        mr = None
        if training:
            stats = torch.empty(2 * H, dtype=torch.float64, device=dev)
            ops.bn_colstats(y, M, H, stats)
            out, mr = _bn_train(y.view(M, H), M, H, bn, stats, False)
        else:
            out = _bn_infer(y.view(M, H), M, H, bn, False)
        ctx.layer, ctx.dims = layer, (B, T, K, H, G, cell)
        ctx.saved = (x2, u, hs_ext, cs_ext, gates, rh, y, mr) if training else None
        return out.view(B, T, H)

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise RuntimeError("recurrent layer: backward needs a training-mode forward")
        layer = ctx.layer
        p, bn = layer.rnn, layer.bn
        B, T, K, H, G, cell = ctx.dims
        x2, u, hs_ext, cs_ext, gates, rh, y, mr = ctx.saved
        ctx.saved = None
        M = B * T
        dev, cd = dout.device, x2.dtype
        _, _, g_in, g_rec = p.bias_views()
        dout = dout.contiguous().view(M, H)
        scratch = ops.bn_relu6_bwd_scratch(M, H, dev)
        dy = torch.empty_like(dout)
        ops.bn_bwd(dout, y, mr, bn.gamma, bn.beta, M, H, scratch, dy, bn.gamma.grad, bn.beta.grad)
        dxw = torch.empty((M, G * H), dtype=cd, device=dev)
        dhu_ext = torch.empty((B, T + 1, G * H), dtype=cd, device=dev)
        ws = ops.rnn_workspace(cell, B, H, dev)
        ops.rnn_bwd(cell, dy, u, hs_ext, cs_ext, gates, rh, dxw, dhu_ext, B, T, H, ws)
        # kernel / input bias from the inputs, recurrent kernel / recurrent bias from the states one step back
        _wgrad(x2, dxw, M, K, G * H, K, G * H, p.kernel.grad, g_in)
        _wgrad(hs_ext.view(B * (T + 1), H), dhu_ext.view(B * (T + 1), G * H), B * (T + 1), H, G * H, H, G * H,
               p.recurrent_kernel.grad, g_rec)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=cd, device=dev)
            if cd == torch.bfloat16:
                ops.gemm(dxw, p.k_shadow, dx, M, K, G * H, trans_b=True, lda=G * H, ldb=p.k_shadow.shape[1], ldc=K)
            else:
                ops.gemm(dxw, p.kernel, dx, M, K, G * H, trans_b=True, lda=G * H, ldb=G * H, ldc=K)
            dx = dx.view(B, T, K)
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class RnnBn(nn.Module):
    def __init__(self, store, index, rnn_type, in_features, hidden):
        super().__init__()
        if rnn_type not in SUPPORTED_RNNS:
            raise KeyError("rnn_type %r: one of %s" % (rnn_type, sorted(SUPPORTED_RNNS)))
        if hidden % 8 or in_features % 8:
            raise ValueError("rnn sizes must be multiples of 8 (got in=%d hidden=%d)" % (in_features, hidden))
        self.store = store
        name = "rnn" if index == 0 else "rnn_%d" % index
        self.rnn = RNNParams(store, name, SUPPORTED_RNNS[rnn_type], in_features, hidden)
        self.bn = BatchNormParams(store, "batch_normalization_%d" % (index + 2), hidden)

    def forward(self, x, training=None):
        return RnnBnFn.apply(x, self, bool(training), *self.parameters())


class DenseActFn(torch.autograd.Function):
    """Dense with an activation fused into the product (deep_speech.py:81)."""

    @staticmethod
    def forward(ctx, x, dense, act, kernel, bias):
        B, T, K = x.shape
        x2 = x.contiguous().view(B * T, K)
        y, _ = lin_fwd(x2, B * T, dense, act=act)
        ctx.dense, ctx.act, ctx.x2, ctx.y, ctx.shape = dense, act, x2, y, (B, T, K)
        return y.view(B, T, dense.out_features)

    @staticmethod
    def backward(ctx, dy):
        dense, act, x2, y = ctx.dense, ctx.act, ctx.x2, ctx.y
        B, T, K = ctx.shape
        M, N = B * T, dense.out_features
        dy = dy.contiguous().view(M, N)
        if act == ACT_RELU6:
            # the relu6 gate is read off the stored output (the library's dropout backward kernel with
            # no dropout applies exactly that gate)
            g = torch.empty_like(dy)
            ops.dropout_bwd(dy, 1.0, 0.0, None, 0, y, g)
            dy = g
        elif act != ACT_NONE:
            raise NotImplementedError("DenseActFn: activation %d" % act)
        lin_bwd_w(x2, dy, M, dense, K, N)
        dx = lin_bwd_x(dy, M, dense, N)
        ctx.x2 = ctx.y = None
        return dx.view(B, T, K), None, None, None, None


__all__ = ["ConvFront", "RnnBn", "DenseActFn", "Dense", "SUPPORTED_RNNS", "ACT_MUL_DRELU6"]
