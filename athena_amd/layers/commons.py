"""Parameter-holding building blocks (mirror of the reference's athena/layers/commons.py:26-41,494-506
for the parts the ASR training step uses)."""
import torch
from torch import nn

from .. import ops
from ..runtime import constant, glorot_uniform

ACTIVATIONS = {"relu6": ops.ACT_RELU6, "gelu": ops.ACT_GELU, "linear": ops.ACT_NONE}

# The activation of the feed-forward block by name (reference ACTIVATIONS, athena/layers/commons.py:494-506):
# name -> (epilogue of the forward product, epilogue of the backward product, which multiplies by the
# derivative at the saved pre-activation).  "glu" halves the width of its input and is not a
# feed-forward activation the transformer layers can take.
_UNARY = ["relu", "elu", "selu", "leaky_relu", "sigmoid", "softplus", "softsign", "tanh", "relu6"]
FFN_ACTIVATIONS = {"gelu": (ops.ACT_GELU, ops.ACT_MUL_DGELU)}
FFN_ACTIVATIONS.update({n: (ops.ACT_UNARY + k, ops.ACT_MUL_DUNARY + k) for k, n in enumerate(_UNARY)})


class _Bound(nn.Module):
    """A module whose variables live in a ParamStore; ``bind()`` creates the Parameter views."""

    def _bind_param(self, attr, name):
        p = nn.Parameter(self.store.view(name), requires_grad=True)
        p.grad = self.store.grad_view(name)
        setattr(self, attr, p)

    def bind(self):
        raise NotImplementedError


class Dense(_Bound):
    """tf.keras.layers.Dense: kernel [in, out], bias [out] (zeros)."""

    def __init__(self, store, name, in_features, out_features, kernel_init=None, tiled=False):
        super().__init__()
        self.store, self.vname = store, name
        self.in_features, self.out_features = in_features, out_features
        store.register(name + ".kernel", (in_features, out_features),
                       kernel_init or glorot_uniform(in_features, out_features), matrix=(in_features, out_features),
                       tiled=tiled)
        store.register(name + ".bias", (out_features,), constant(0.0))
        self.shadow = self.shadow_t = self.tiled = self.tiled_t = None

    def bind(self):
        self._bind_param("kernel", self.vname + ".kernel")
        self._bind_param("bias", self.vname + ".bias")
        if self.store.shadow is not None:
            self.shadow, self.shadow_t = self.store.shadow_views(self.vname + ".kernel")
            self.tiled, self.tiled_t = self.store.tiled_views(self.vname + ".kernel")


class Conv3x3Params(_Bound):
    """Kernel of a Conv2D(3x3, stride 2, same, no bias): HWIO [3,3,Cin,Cout], glorot-uniform."""

    def __init__(self, store, name, cin, cout, as_matrix):
        super().__init__()
        self.store, self.vname, self.cin, self.cout = store, name, cin, cout
        store.register(name + ".kernel", (3, 3, cin, cout), glorot_uniform(9 * cin, 9 * cout),
                       matrix=(9 * cin, cout) if as_matrix else None)
        self.as_matrix = as_matrix
        self.shadow = self.shadow_t = None

    def bind(self):
        self._bind_param("kernel", self.vname + ".kernel")
        if self.as_matrix and self.store.shadow is not None:
            self.shadow, self.shadow_t = self.store.shadow_views(self.vname + ".kernel")


class LayerNormParams(_Bound):
    """gamma / beta of tf.keras.layers.LayerNormalization."""

    def __init__(self, store, name, d):
        super().__init__()
        self.store, self.vname = store, name
        store.register(name + ".gamma", (d,), constant(1.0))
        store.register(name + ".beta", (d,), constant(0.0))

    def bind(self):
        self._bind_param("gamma", self.vname + ".gamma")
        self._bind_param("beta", self.vname + ".beta")


class BatchNormParams(_Bound):
    """gamma / beta / moving statistics of tf.keras.layers.BatchNormalization()."""

    def __init__(self, store, name, c):
        super().__init__()
        self.store, self.vname = store, name
        store.register(name + ".gamma", (c,), constant(1.0))
        store.register(name + ".beta", (c,), constant(0.0))
        store.register(name + ".moving_mean", (c,), constant(0.0), trainable=False)
        store.register(name + ".moving_variance", (c,), constant(1.0), trainable=False)

    def bind(self):
        self._bind_param("gamma", self.vname + ".gamma")
        self._bind_param("beta", self.vname + ".beta")
        self.moving_mean = self.store.view(self.vname + ".moving_mean")
        self.moving_variance = self.store.view(self.vname + ".moving_variance")


class PositionalEncoding(nn.Module):
    """x * sqrt(d) (if scale) + sinusoid table (reference athena/layers/commons.py:26-41)."""

    def __init__(self, store, d_model, max_position=800, scale=False):
        super().__init__()
        self.store, self.d_model, self.max_position, self.scale = store, d_model, max_position, scale

    def forward(self, x):
        B, T, d = x.shape
        if T > self.max_position:
            raise ValueError("sequence length %d exceeds max_position %d" % (T, self.max_position))
        y = torch.empty_like(x)
        xs = float(torch.sqrt(torch.tensor(float(d), dtype=torch.float32))) if self.scale else 1.0
        ops.add_pe_dropout(x.contiguous(), self.store.pe(d, self.max_position), B, T, d, xs, 0.0, None, 0, y)
        return y


def bind_all(module):
    """Create the Parameter views of every bound sub-module (call after ParamStore.finalize)."""
    for m in module.modules():
        if isinstance(m, _Bound):
            m.bind()
