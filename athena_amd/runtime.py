"""Device-side state shared by the layers: flat parameter storage, bf16 weight shadows,
the dropout RNG state and the positional-encoding table.

Memory layout in HBM (one GPU, one process):
  * ``params`` / ``grads`` / Adam ``m`` / ``v``: four flat fp32 buffers with identical layout, one
    256-byte aligned slot per variable, in registration (= Keras variable) order.  Every
    ``nn.Parameter`` is a view into ``params`` and its ``.grad`` a view into ``grads``; the
    optimizer and the gradient all-reduce work on the flat buffers.
  * ``shadow``: one flat bf16 buffer holding, for every 2-D weight [in, out], a row-major copy
    (leading dimension rounded up to 8) and a transposed copy [out, in]; refreshed by one kernel
    after each optimizer step.  Only the bf16 precision mode allocates it.
"""
import math

import numpy as np
import torch

from . import ops

ALIGN = 64   # elements (256 B for fp32)


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("athena_amd needs a ROCm GPU: torch.cuda.is_available() is False and there is "
                           "no CPU fallback for the compute path")
    return torch.device("cuda", torch.cuda.current_device())


def make_positional_encoding(position, d_model):
    """Sinusoid table [position, d_model] in fp32 (reference athena/layers/functional.py:25-38)."""
    pos = np.arange(position)[:, np.newaxis]
    i = np.arange(d_model)[np.newaxis, :]
    angle_rates = 1 / np.power(10000, (2 * (i // 2)) / np.float32(d_model))
    ang = pos * angle_rates
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(ang.astype(np.float32))


class Slot:
    """One registered variable: name, shape, offset into the flat buffers."""
    __slots__ = ("name", "shape", "numel", "offset", "trainable", "matrix", "sh_off", "sh_ld", "sht_off",
                 "sht_ld", "init", "group", "tiled", "tl_off", "tlt_off")

    def __init__(self, name, shape, trainable, matrix, init, tiled=False):
        self.name, self.shape, self.trainable, self.matrix, self.init = name, tuple(shape), trainable, matrix, init
        self.numel = int(np.prod(shape))
        self.offset = self.sh_off = self.sh_ld = self.sht_off = self.sht_ld = self.group = None
        self.tiled, self.tl_off, self.tlt_off = tiled, -1, -1


class ParamStore:
    """Flat storage for all variables of one model (see module docstring)."""

    def __init__(self, compute_dtype=torch.bfloat16, device=None):
        self.compute_dtype = compute_dtype
        self.device = device
        self.slots = []
        self.by_name = {}
        self.finalized = False
        self.shadows_dirty = True
        # activations at which the backward pass may be cut into pieces (filled during a forward pass
        # when ``collect_cuts`` is set; see solver._PipelinedGraphStep): list of (tag, tensor)
        self.collect_cuts = False
        self.cuts = []
        self._site = 0
        self.shadow_groups = []
        self.shadow_stacks = []
        self.stacks = {}

    # ---- registration (before finalize) -------------------------------------------------
    def register(self, name, shape, init, trainable=True, matrix=None, tiled=False):
        """``matrix`` = (rows, cols) if the variable is used as a GEMM operand (gets bf16 shadows);
        ``tiled`` asks for the two fragment-tiled copies as well (``tiled_views``; given when both
        dimensions are multiples of 32)."""
        if self.finalized:
            raise RuntimeError("ParamStore: register after finalize")
        if name in self.by_name:
            raise KeyError("ParamStore: duplicate variable %s" % name)
        s = Slot(name, shape, trainable, matrix, init, tiled)
        self.slots.append(s)
        self.by_name[name] = s
        return s

    def group_shadows(self, names):
        """Lay the bf16 shadows of several matrices [in, out_i] with equal ``in`` side by side: the
        row-major copies form one [in, sum out_i] matrix and the transposed copies one
        [sum out_i, in] matrix, so products with all of them can run as a single GEMM (each member
        stays addressable as a column block / row block of the fused copy)."""
        if self.finalized:
            raise RuntimeError("ParamStore: group_shadows after finalize")
        self.shadow_groups.append(list(names))

    def stack_shadows(self, tag, names):
        """One MORE row-major bf16 copy of several matrices [in, out_i] with equal ``in``, side by side as
        one [in, sum out_i] matrix (``stack_view(tag)``), whatever shadow groups the members belong to:
        the gradients of one input through several Dense layers -- the decoder layers' cross-attention
        key / value projections of the memory -- are then ONE product with their output gradients laid
        side by side."""
        if self.finalized:
            raise RuntimeError("ParamStore: stack_shadows after finalize")
        self.shadow_stacks.append((tag, list(names)))

    def stack_view(self, tag):
        """The stacked copy [in, sum out_i] of ``stack_shadows(tag, ...)``, or None (fp32 mode, unknown tag)."""
        st = self.stacks.get(tag)
        if st is None or self.shadow is None:
            return None
        base, r, total_c = st
        return self.shadow[base:base + r * total_c].view(r, total_c)

    def new_site(self):
        """A fresh dropout site id (distinct RNG stream per dropout layer)."""
        self._site += 1
        return self._site

    # ---- allocation ------------------------------------------------------------------------
    def finalize(self, seed=1):
        dev = self.device or default_device()
        self.device = dev
        # trainables first so the optimizer's flat range is contiguous
        order = [s for s in self.slots if s.trainable] + [s for s in self.slots if not s.trainable]
        off = 0
        for s in order:
            s.offset = off
            off += ops.roundup(s.numel, ALIGN)
            if s.trainable:
                self.n_train = off
        self.total = off
        self.train_slots = [s for s in order if s.trainable]
        host = torch.zeros(self.total, dtype=torch.float32)
        gen = torch.Generator().manual_seed(seed)
        for s in self.slots:     # registration order, so a seed reproduces across precisions
            host[s.offset:s.offset + s.numel] = s.init(s.shape, gen).reshape(-1).to(torch.float32)
        self.params = host.to(dev)
        self.grads = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.seg_offsets = torch.tensor([s.offset for s in self.train_slots] + [self.n_train],
                                        dtype=torch.int64, device=dev)
        # bf16 shadows
        self.shadow = None
        if self.compute_dtype == torch.bfloat16:
            soff = 0
            table, prefix = [], [0]
            grouped = {}
            for names in self.shadow_groups:
                members = [self.by_name[n] for n in names]
                r = members[0].matrix[0]
                if any(m.matrix is None or m.matrix[0] != r or m.matrix[1] % 8 or r % 8 for m in members):
                    raise ValueError("ParamStore: shadows %s cannot be grouped" % (names,))
                total_c = sum(m.matrix[1] for m in members)
                col = 0
                for m in members:
                    grouped[m.name] = (names[0], col, total_c)
                    col += m.matrix[1]
            group_base = {}
            for s in self.slots:
                if s.matrix is None:
                    continue
                r, c = s.matrix
                if s.name in grouped:
                    first, col, total_c = grouped[s.name]
                    if first not in group_base:
                        # members of a group are registered one after the other; space for all of them
                        # is taken when the first one comes by
                        group_base[first] = (soff, soff + r * total_c)
                        soff += ops.roundup(2 * r * total_c, ALIGN)
                    base, base_t = group_base[first]
                    s.sh_ld, s.sht_ld = total_c, r
                    s.sh_off = base + col
                    s.sht_off = base_t + col * r
                    s.group = (base, base_t, total_c)
                else:
                    s.sh_ld = ops.roundup(c, 8)
                    s.sht_ld = ops.roundup(r, 8)
                    s.sh_off = soff
                    soff += ops.roundup(r * s.sh_ld, ALIGN)
                    s.sht_off = soff
                    soff += ops.roundup(c * s.sht_ld, ALIGN)
                if s.tiled and r % 32 == 0 and c % 32 == 0 and s.sh_ld % 4 == 0 and s.sht_ld % 4 == 0:
                    s.tl_off = soff
                    s.tlt_off = soff + r * c
                    soff += ops.roundup(2 * r * c, ALIGN)
                table += [s.offset, r, c, s.sh_off, s.sh_ld, s.sht_off, s.sht_ld, s.tl_off, s.tlt_off]
                prefix.append(prefix[-1] + ((r + 31) // 32) * ((c + 31) // 32))
            for tag, names in self.shadow_stacks:
                members = [self.by_name[n] for n in names]
                rr = members[0].matrix[0]
                if any(m.matrix is None or m.matrix[0] != rr or m.matrix[1] % 8 or rr % 8 for m in members):
                    raise ValueError("ParamStore: shadows %s cannot be stacked" % (names,))
                total_c = sum(m.matrix[1] for m in members)
                base = soff
                soff += ops.roundup(rr * total_c, ALIGN)
                col = 0
                for m in members:
                    # (an entry always writes a transposed copy too: the member's own, once more)
                    table += [m.offset, rr, m.matrix[1], base + col, total_c, m.sht_off, m.sht_ld, -1, -1]
                    prefix.append(prefix[-1] + ((rr + 31) // 32) * ((m.matrix[1] + 31) // 32))
                    col += m.matrix[1]
                self.stacks[tag] = (base, rr, total_c)
            self.shadow = torch.zeros(max(soff, 8), dtype=torch.bfloat16, device=dev)
            self.sh_table = torch.tensor(table, dtype=torch.int64, device=dev)
            self.sh_prefix = torch.tensor(prefix, dtype=torch.int64, device=dev)
            self.sh_nmat = len(prefix) - 1
            self.sh_tiles = prefix[-1]
        # dropout RNG state {seed, step}
        self.rng_state = torch.tensor([seed * 0x9E3779B97F4A7C15 % (1 << 63), 0], dtype=torch.int64, device=dev)
        self.pe_cache = {}
        self.finalized = True
        self.shadows_dirty = True

    # ---- views ---------------------------------------------------------------------------------
    def view(self, name):
        s = self.by_name[name]
        return self.params[s.offset:s.offset + s.numel].view(s.shape)

    def grad_view(self, name):
        s = self.by_name[name]
        return self.grads[s.offset:s.offset + s.numel].view(s.shape)

    def shadow_views(self, name):
        """(row-major bf16 copy [rows, ld], transposed copy [cols, ld_t]) of a matrix variable."""
        s = self.by_name[name]
        r, c = s.matrix
        a = self.shadow[s.sh_off:s.sh_off + r * s.sh_ld].view(r, s.sh_ld)
        b = self.shadow[s.sht_off:s.sht_off + c * s.sht_ld].view(c, s.sht_ld)
        return a, b

    def tiled_views(self, name):
        """(tiled copy with the matrix' rows as rows, tiled copy with its columns as rows) -- flat bf16
        tensors in the block layout ``athena_refresh_shadows`` documents -- or (None, None)."""
        s = self.by_name[name]
        if s.tl_off < 0:
            return None, None
        n = s.matrix[0] * s.matrix[1]
        return self.shadow[s.tl_off:s.tl_off + n], self.shadow[s.tlt_off:s.tlt_off + n]

    def group_shadow_views(self, name):
        """(fused row-major copy [in, sum out_i], fused transposed copy [sum out_i, in]) of the
        shadow group the matrix ``name`` belongs to."""
        s = self.by_name[name]
        base, base_t, total_c = s.group
        r = s.matrix[0]
        return (self.shadow[base:base + r * total_c].view(r, total_c),
                self.shadow[base_t:base_t + r * total_c].view(total_c, r))

    def refresh_shadows(self, force=False):
        if self.shadow is None or self.sh_nmat == 0:
            self.shadows_dirty = False
            return
        if self.shadows_dirty or force:
            ops.refresh_shadows(self.params, self.shadow, self.sh_table, self.sh_nmat, self.sh_prefix,
                                self.sh_tiles)
            self.shadows_dirty = False

    def pe(self, d_model, max_position=800):
        key = (d_model, max_position)
        if key not in self.pe_cache:
            self.pe_cache[key] = make_positional_encoding(max_position, d_model).to(self.device)
        return self.pe_cache[key]

    # ---- import / export (flat name -> fp32 tensor, Keras variable shapes) ------------------
    def state_dict(self):
        return {s.name: self.view(s.name).detach().cpu().clone() for s in self.slots}

    def load_state_dict(self, state, strict=True):
        for s in self.slots:
            if s.name not in state:
                if strict:
                    raise KeyError("missing variable %s" % s.name)
                continue
            t = torch.as_tensor(state[s.name]).to(torch.float32)
            if tuple(t.shape) != s.shape:
                raise ValueError("variable %s: shape %s, expected %s" % (s.name, tuple(t.shape), s.shape))
            self.view(s.name).copy_(t.to(self.device))
        self.shadows_dirty = True


# ---- initialisers (Keras defaults restated) --------------------------------------------------------
def glorot_uniform(fan_in, fan_out):
    def f(shape, gen):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim
    return f


def truncated_normal(std=0.02):
    def f(shape, gen):
        t = torch.empty(shape, dtype=torch.float64)
        torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=gen)
        return t
    return f


def uniform(lo, hi):
    def f(shape, gen):
        return torch.rand(shape, generator=gen, dtype=torch.float64) * (hi - lo) + lo
    return f


def constant(v):
    def f(shape, gen):
        return torch.full(shape, float(v), dtype=torch.float64)
    return f
