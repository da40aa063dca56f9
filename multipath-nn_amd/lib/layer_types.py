"""Layer types: the operator surface of multipath-nn, MI355X-native.

Mirrors the class names, constructor keywords and attributes of the reference
module ``scripts/lib/layer_types.py`` so that experiment specs written against
it (``scripts/arch_and_hypers.py``) import and run unchanged.  What differs is
what ``link`` does: the reference appends TensorFlow ops to the default graph;
here ``link(x, y, mode)`` performs shape inference, registers parameters with
the net being linked (``_linker``) and records the cost terms (``n_ops``, L2
terms).  The arithmetic itself is compiled by ``lib/_plan.py`` into launches
of the hand-written HIP kernels in ``csrc/`` -- there is no CPU fallback.

Reference protocol (scripts/lib/layer_types.py:11-26): the constructor pops
``name, router, sinks, comps`` and folds everything else into ``hypers`` over
the class's ``default_hypers``; ``link`` sets ``x, c_err, c_mod, n_ops`` and,
for error layers, ``δ_cor``.
"""
from abc import ABCMeta
from types import SimpleNamespace as Ns

import numpy as np

################################################################################
# Symbolic activations, parameters and the linker context
################################################################################

class Sym:
    """A symbolic activation: per-sample shape plus, once a plan is compiled,
    the device buffers that hold it (``buf``: pre-activation values;
    ``bn``: the BatchNorm whose normalise+ReLU is applied on load, or None)."""

    def __init__(self, shape, producer=None):
        self.shape = tuple(int(s) for s in shape)
        self.producer = producer
        self.buf = None
        self.bn = None
        self.relu = False
        self.shift = 0          # log2 subsampling applied on load (ToPyramid)

    @property
    def n_el(self):
        return int(np.prod(self.shape))

    def __repr__(self):
        return 'Sym%s' % (self.shape,)


class Param:
    """One parameter tensor.  ``init`` = (kind, scale, eq) with kind in
    {'normal', 'zeros', 'ones'}; value = eq + scale * N(0, 1) for 'normal'.
    After ``Net`` allocation ``data``/``grad``/``accum`` are views into the
    flat device buffers (trainable) or the state buffer (``trainable=False``)."""

    def __init__(self, name, shape, init, trainable=True, l2=0.0, eq=None):
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self.init = init
        self.trainable = trainable
        self.l2 = float(l2)         # k_l2 coefficient of k_l2 * sum((w - eq)^2)
        self.eq = eq                # None or ndarray (residual init)
        self.owner = None
        self.offset = None
        self.data = None
        self.grad = None
        self.accum = None

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    def numpy(self):
        return self.data.detach().cpu().numpy().reshape(self.shape)

    eval = numpy                    # reference idiom: ``v.eval()`` (serdes.py:16)

    def assign(self, value):
        import torch
        v = torch.as_tensor(np.asarray(value, np.float32).reshape(-1))
        self.data.copy_(v.to(self.data.device))
        hook = getattr(self, '_on_assign', None)
        if hook is not None:                       # (the engine rebuilds its weight packs before the next run)
            hook()

    def __repr__(self):
        return 'Param(%s%s)' % (self.name, self.shape)


class _Linker:
    """Stands in for TensorFlow's default graph while a net is being linked."""
    stack = []

    def __init__(self):
        self.params = []

    def __enter__(self):
        _Linker.stack.append(self)
        return self

    def __exit__(self, *exc):
        _Linker.stack.pop()

    @staticmethod
    def register(layer, p):
        p.owner = layer
        if _Linker.stack:
            _Linker.stack[-1].params.append(p)
        return p


def _shape(x):
    return list(x.shape)

################################################################################
# Core Layer Class  (reference: scripts/lib/layer_types.py:11-26)
################################################################################

class Layer(metaclass=ABCMeta):
    default_hypers = Ns()

    def __init__(self, **options):
        self.name = options.pop('name', type(self).__name__)
        self.router = options.pop('router', None)
        self.sinks = list(options.pop('sinks', []))
        self.comps = list(options.pop('comps', []))
        self.hypers = Ns(**{**vars(type(self).default_hypers), **options})
        self.params = Ns()

    def link(self, x, y, mode):
        self.x = x
        self.c_err = 0.0
        self.c_mod = 0.0
        self.n_ops = 0
        self.l2_terms = []

    def _param(self, key, shape, init, trainable=True, l2=0.0, eq=None):
        p = _Linker.register(self, Param(key, shape, init, trainable, l2, eq))
        setattr(self.params, key, p)
        if l2:
            self.l2_terms.append(p)
        return p

################################################################################
# The No-Op Layer
################################################################################

class NoOp(Layer):
    pass

################################################################################
# Transformation Layers
################################################################################

class LinTrans(Layer):
    """Flatten + affine map.  Reference: scripts/lib/layer_types.py:39-53."""
    default_hypers = Ns(n_chan=1, k_l2=0, σ_w=1, res=False)

    def link(self, x, y, mode):
        super().link(x, y, mode)
        ϕ = self.hypers
        n_in = int(np.prod(_shape(x)))
        w_eq = np.eye(n_in, ϕ.n_chan, dtype=np.float32) if ϕ.res else None
        self._param('w', (n_in, ϕ.n_chan),
                    ('normal', ϕ.σ_w / np.sqrt(n_in)), l2=ϕ.k_l2, eq=w_eq)
        self._param('b', (ϕ.n_chan,), ('zeros', 0.0))
        self.x = Sym((ϕ.n_chan,), self)
        self.n_ops = n_in * ϕ.n_chan


class Conv(Layer):
    """Single-scale SAME convolution.  Reference: layer_types.py:55-74."""
    default_hypers = Ns(n_chan=1, supp=1, k_l2=0, σ_w=1, res=False)

    def link(self, x, y, mode):
        super().link(x, y, mode)
        ϕ = self.hypers
        h, w, n_in = _shape(x)
        w_eq = None
        if ϕ.res:
            mid = (np.arange(ϕ.supp) == ϕ.supp // 2)
            w_eq = np.float32(mid[:, None, None, None] * mid[None, :, None, None]
                              * np.eye(n_in, ϕ.n_chan))
        self._param('w', (ϕ.supp, ϕ.supp, n_in, ϕ.n_chan),
                    ('normal', ϕ.σ_w / ϕ.supp / np.sqrt(n_in)), l2=ϕ.k_l2, eq=w_eq)
        self._param('b', (ϕ.n_chan,), ('zeros', 0.0))
        self.x = Sym((h, w, ϕ.n_chan), self)
        self.n_ops = h * w * ϕ.supp ** 2 * n_in * ϕ.n_chan


class Rect(Layer):
    """ReLU.  Reference: layer_types.py:76-79."""

    def link(self, x, y, mode):
        super().link(x, y, mode)
        self.x = Sym(x.shape, self)


class Softmax(Layer):
    """Row softmax.  Reference: layer_types.py:81-84."""

    def link(self, x, y, mode):
        super().link(x, y, mode)
        self.x = Sym(x.shape, self)


class MaxPool(Layer):
    """Reference: layer_types.py:86-94 (unused by every shipped spec).  The reference calls
    ``tf.nn.max_pool(x, strides, k_shape, 'SAME')`` where TensorFlow's signature is ``(value, ksize, strides,
    padding)``: its window is ``hypers.stride`` and its step ``hypers.supp``.  Kept, so that the same spec gives the
    same shapes: the output is ceil(H / supp) x ceil(W / supp).  Runs in a single-scale Conv chain (lib/_plan_conv.py,
    csrc/pool.hip)."""
    default_hypers = Ns(stride=1, supp=1)

    def link(self, x, y, mode):
        super().link(x, y, mode)
        h, w, c = _shape(x)
        s = self.hypers.supp
        self.x = Sym((-(-h // s), -(-w // s), c), self)


class GlobalMaxPool(Layer):
    """Reference: layer_types.py:96-100 (unused by every shipped spec): tf.reduce_max over the spatial dims.  Runs in a
    single-scale Conv chain (lib/_plan_conv.py, csrc/pool.hip)."""

    def link(self, x, y, mode):
        super().link(x, y, mode)
        self.x = Sym(x.shape[-1:], self)

################################################################################
# Multiscale Transformation Layers
################################################################################

def n_pix(x):
    return int(np.prod(x.shape[:2]))


class ToPyramid(Layer):
    """n_scales copies of the input, scale i subsampled by 2**i.
    Reference: layer_types.py:118-125 (legacy bilinear resize at an integer
    ratio = strided pick; folded into the first conv's load addressing)."""
    default_hypers = Ns(n_scales=1)

    def link(self, x, y, mode):
        super().link(x, y, mode)
        h, w, c = _shape(x)
        self.x = []
        for i in range(self.hypers.n_scales):
            s = Sym((h // 2 ** i, w // 2 ** i, c), self)
            s.shift = i
            self.x.append(s)


class MultiscaleLLN(Layer):
    """Imported by arch_and_hypers.py but never instantiated by any spec
    (reference: layer_types.py:127-147).  Outside the hot path."""
    default_hypers = Ns(shape0=(1, 1), σ=3, ϵ=1e-3)

    def link(self, x, y, mode):
        raise NotImplementedError(
            'MultiscaleLLN is outside the MI355X hot path (no shipped spec uses it)')


class MultiscaleConvMax(Layer):
    """out[0] = b0 + conv(x[-L], wh0); out[i] = bi + conv(x[i], wh_i)
    + conv(maxpool2(out[i-1]), wv_i).  Reference: layer_types.py:149-194."""
    default_hypers = Ns(n_chan=[], supp=1, k_l2=0, σ_w=1)

    def link(self, x, y, mode):
        super().link(x, y, mode)
        ϕ = self.hypers
        L = len(ϕ.n_chan)
        xs = list(x)[-L:]                       # negative indexing in the reference
        self.x_in = xs
        self.x = []
        self.n_ops = 0
        for i, x_i in enumerate(xs):
            h, w, n_in = _shape(x_i)
            kh, kw = min(ϕ.supp, h), min(ϕ.supp, w)
            self._param('w_horz_%i' % i, (kh, kw, n_in, ϕ.n_chan[i]),
                        ('normal', ϕ.σ_w / ϕ.supp / np.sqrt(n_in)), l2=ϕ.k_l2)
        for i in range(L - 1):
            self._param('w_vert_%i' % i, (ϕ.supp, ϕ.supp, ϕ.n_chan[i], ϕ.n_chan[i + 1]),
                        ('normal', ϕ.σ_w / ϕ.supp / np.sqrt(ϕ.n_chan[i])), l2=ϕ.k_l2)
        for i in range(L):
            self._param('b_%i' % i, (ϕ.n_chan[i],), ('zeros', 0.0))
        for i, x_i in enumerate(xs):
            h, w, _ = _shape(x_i)
            self.x.append(Sym((h, w, ϕ.n_chan[i]), self))
            self.n_ops += h * w * (
                getattr(self.params, 'w_horz_%i' % i).size
                + (getattr(self.params, 'w_vert_%i' % (i - 1)).size if i > 0 else 0))


class MultiscaleRect(Layer):
    """Reference: layer_types.py:196-199."""

    def link(self, x, y, mode):
        super().link(x, y, mode)
        self.x = [Sym(x_i.shape, self) for x_i in x]


class Select(Layer):
    """Reference: layer_types.py:201-206."""
    default_hypers = Ns(i=0)

    def link(self, x, y, mode):
        super().link(x, y, mode)
        self.x = x[self.hypers.i]

################################################################################
# Regularization Layers
################################################################################

class Dropout(Layer):
    """Reference: layer_types.py:212-217 (unused by every shipped spec)."""
    default_hypers = Ns(λ=1)

    def link(self, x, y, mode):
        raise NotImplementedError('Dropout is outside the MI355X hot path')


class BatchNorm(Layer):
    """Per-channel normalisation over all leading dims; batch statistics in
    'tr' (biased variance, moving averages updated with decay d), moving
    averages otherwise.  Reference: layer_types.py:219-239."""
    default_hypers = Ns(d=0.9, ϵ=1e-6)

    def link(self, x, y, mode):
        super().link(x, y, mode)
        n_chan = x.shape[-1]
        self._param('γ', (n_chan,), ('ones', 0.0))
        self._param('β', (n_chan,), ('zeros', 0.0))
        self._param('m_avg', (n_chan,), ('zeros', 0.0), trainable=False)
        self._param('v_avg', (n_chan,), ('ones', 0.0), trainable=False)
        self.x = Sym(x.shape, self)


class MultiscaleBatchNorm(Layer):
    """One independent BatchNorm per scale.  Reference: layer_types.py:241-249.

    The reference accepts `d` / `ϵ` here and DISCARDS them: every scale gets a
    `BatchNorm()` with the default hypers (layer_types.py:246).  Same here, so
    a spec that passes either gives the reference's numbers, not the numbers
    its author may have meant."""
    default_hypers = Ns(d=0.9, ϵ=1e-6)

    def link(self, x, y, mode):
        super().link(x, y, mode)
        self.comps = [BatchNorm() for _ in x]
        for ℓ, x_i in zip(self.comps, x):
            ℓ.link(x_i, y, mode)
        self.x = [ℓ.x for ℓ in self.comps]

################################################################################
# Error Layers
################################################################################

class CrossEntropyError(Layer):
    """c_err = -sum(y * log(ϵ/n_cls + (1-ϵ) x)); δ_cor = [argmax x == argmax y].
    Reference: layer_types.py:262-272."""
    default_hypers = Ns(ϵ=1e-6)

    def link(self, x, y, mode):
        super().link(x, y, mode)
        self.c_err = Sym((), self)
        self.δ_cor = Sym((), self)


class SquaredError(Layer):
    """Reference: layer_types.py:255-260 (unused by every shipped spec)."""

    def link(self, x, y, mode):
        raise NotImplementedError('SquaredError is outside the MI355X hot path')


class SuperclassCrossEntropyError(Layer):
    """Reference: layer_types.py:274-285 (unused by every shipped spec)."""
    default_hypers = Ns(w_cls=None, ϵ=1e-6)

    def link(self, x, y, mode):
        raise NotImplementedError(
            'SuperclassCrossEntropyError is outside the MI355X hot path')


class ActivityError(Layer):
    """Reference: layer_types.py:287-293 (unused by every shipped spec)."""
    default_hypers = Ns(α=0.0)

    def link(self, x, y, mode):
        raise NotImplementedError('ActivityError is outside the MI355X hot path')

################################################################################
# Compound Layers
################################################################################

class Chain(Layer):
    """Sequential composite; sums costs and op counts, forwards the last
    component's δ_cor.  Reference: layer_types.py:299-310."""

    def link(self, x, y, mode):
        super().link(x, y, mode)
        for ℓ in self.comps:
            ℓ.link(x, y, mode)
            x = ℓ.x
        self.x = x
        errs = [ℓ.c_err for ℓ in self.comps if isinstance(ℓ.c_err, Sym)]
        self.c_err = errs[-1] if errs else 0.0
        self.n_ops = sum(ℓ.n_ops for ℓ in self.comps)
        self.l2_terms = [p for ℓ in self.comps for p in ℓ.l2_terms]
        if len(self.comps) > 0 and hasattr(self.comps[-1], 'δ_cor'):
            self.δ_cor = self.comps[-1].δ_cor
