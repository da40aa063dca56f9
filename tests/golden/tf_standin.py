"""A stand-in for the pre-1.0 TensorFlow API surface that the reference's graph-assembly code uses
(scripts/lib/layer_types.py, scripts/lib/net_types.py), evaluated with float64 torch on the CPU.

FIXTURE TOOLING, used only by tests/golden/make_ref_graph_golden.py in the build container.  It lets
the REFERENCE'S OWN Python (Layer.link, Net.link, _route*, the cost assembly, minimize_expectation)
run unmodified and produce golden vectors, so that the restatement in oracle/ref_net.py is checked
against the reference's code instead of against a reading of it.

WHAT THIS DOES NOT DO: pin TensorFlow's operator semantics.  conv2d, max_pool (+ its gradient's
tie-break), moments, argmax, resize_images, MomentumOptimizer are implemented HERE, from the same
assumptions as oracle/np_ops.py -- a stand-in library is not TensorFlow, and parity
with the TensorFlow reference stays unpinned.  What the fixtures pin is everything the reference
expresses in Python on top of those operators: which scales feed which conv, the parameter naming,
the routing probabilities with their epsilon floors, hard routing, c_ev / c_opt / c_cre, the
stop-gradients, the cost assembly, the per-node TALR scales and the router factor, the k_cpt column.

Design: every op evaluates EAGERLY on dummy placeholder values while the graph is built (that gives
static shapes: get_shape()) and records a closure; Session-style `run(fetches, feed)` re-evaluates the
closures lazily and memoised with the fed values.  Variables are torch leaves; gradients come from
torch.autograd on the re-evaluated graph; `assign` only takes effect in `run`.
"""
import contextlib

import numpy as np
import torch

float32 = 'float32'
F = torch.float64
_BATCH = 2            # dummy batch used while the graph is built
_ctrl_stack = []
_rng = np.random.RandomState(0)


class _Dim:
    def __init__(self, v):
        self.value = v


class _Shape:
    def __init__(self, dims):
        self.dims = list(dims)

    def as_list(self):
        return list(self.dims)

    def __len__(self):
        return len(self.dims)

    def __getitem__(self, i):
        return _Dim(self.dims[i])


class T:
    """A graph node: `fn(ev)` computes its value given an evaluator ev(node) -> value."""

    def __init__(self, fn, dynamic_batch=True, ctrl=None):
        self.fn = fn
        self.ctrl = list(ctrl if ctrl is not None else (_ctrl_stack[-1] if _ctrl_stack else []))
        self.dynamic_batch = dynamic_batch
        self.value = self._eval_build()

    def _eval_build(self):
        return self.fn(lambda n: n.value if isinstance(n, T) else n)

    def get_shape(self):
        v = self.value
        dims = list(v.shape) if hasattr(v, 'shape') else []
        if dims and self.dynamic_batch and dims[0] == _BATCH:
            dims[0] = None
        return _Shape(dims)

    # arithmetic (ndarray + T must reach T.__radd__ as it reaches a TensorFlow tensor's: `np.eye(..) + w_scale * tf.random_normal(..)`,
    # layer_types.py:49 with res=True)
    __array_ufunc__ = None

    def _bin(self, other, op, rev=False):
        other = as_T(other)
        a, b = (other, self) if rev else (self, other)
        return T(lambda ev: op(_t(ev(a)), _t(ev(b))))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add, True)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return self._bin(o, torch.sub, True)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul, True)
    def __truediv__(self, o): return self._bin(o, torch.div)
    def __rtruediv__(self, o): return self._bin(o, torch.div, True)
    def __neg__(self): return T(lambda ev: -_t(ev(self)))

    def __getitem__(self, idx):
        return T(lambda ev: _t(ev(self))[idx])

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other

    # session idioms
    def run(self, feed=None):
        return run(self, feed or {})

    def eval(self, feed=None):
        return run(self, feed or {})


def _t(v):
    if isinstance(v, torch.Tensor):
        return v
    if isinstance(v, str):
        return v
    return torch.as_tensor(np.asarray(v, np.float64), dtype=F)


def as_T(x):
    if isinstance(x, T):
        return x
    if isinstance(x, (list, tuple)) and any(isinstance(e, T) for e in x):
        elems = [as_T(e) for e in x]
        return T(lambda ev: torch.stack([_t(ev(e)).reshape(()) for e in elems]))
    return T(lambda ev: _t(x), dynamic_batch=False)


class Variable(T):
    def __init__(self, initial_value, trainable=True):
        init = initial_value.value if isinstance(initial_value, T) else _t(initial_value)
        self.data = init.detach().clone().to(F).requires_grad_(bool(trainable))
        self.trainable = trainable
        super().__init__(lambda ev: self.data, dynamic_batch=False, ctrl=[])
        _all_variables.append(self)

    def load(self, value):
        self.data = torch.as_tensor(np.asarray(value, np.float64), dtype=F).reshape(self.data.shape).clone().requires_grad_(self.trainable)
        self.value = self.data


_all_variables = []


class _Placeholder(T):
    def __init__(self, shape, default=None, string=False):
        self.default = default
        if default is not None:
            dummy = default if isinstance(default, str) else _t(default)
        else:
            dummy = torch.zeros([_BATCH if d is None else d for d in shape], dtype=F)
        self.dummy = dummy
        super().__init__(lambda ev: self.dummy, ctrl=[])


def placeholder(dtype, shape=None):
    return _Placeholder(tuple(shape))


def placeholder_with_default(default, shape):
    return _Placeholder(tuple(shape), default=default)


# ---------------------------------------------------------------------------------- evaluation
class _Run:
    def __init__(self, feed):
        self.feed = feed
        self.memo = {}
        self.assigns = []

    def ev(self, n):
        if not isinstance(n, T):
            return n
        k = id(n)
        if k in self.memo:
            return self.memo[k]
        for c in n.ctrl:
            self.ev(c)
        if isinstance(n, _Placeholder):
            if n in self.feed:
                v = self.feed[n]
                v = v if isinstance(v, str) else _t(v)
            elif n.default is not None:
                v = n.dummy
            else:
                raise KeyError('placeholder not fed')
        else:
            self.current = self
            _active.append(self)
            try:
                v = n.fn(self.ev)
            finally:
                _active.pop()
        self.memo[k] = v
        return v


_active = []


def run(fetches, feed):
    r = _Run(feed)
    if isinstance(fetches, dict):
        out = {k: r.ev(v) for k, v in fetches.items()}
    elif isinstance(fetches, (list, tuple)):
        out = [r.ev(v) for v in fetches]
    else:
        out = r.ev(fetches)
    for var, val in r.assigns:                  # assignments take effect when the run is over
        var.data = val.detach().clone().requires_grad_(var.trainable)
        var.value = var.data
    conv = lambda v: v.detach().numpy() if isinstance(v, torch.Tensor) else v
    if isinstance(out, dict):
        return {k: conv(v) for k, v in out.items()}
    if isinstance(out, list):
        return [conv(v) for v in out]
    return conv(out)


# ---------------------------------------------------------------------------------- ops
def _resolve_shape(shape, ev):
    return [int(ev(s)) if isinstance(s, T) else (int(s) if s is not None else -1) for s in (shape if isinstance(shape, (list, tuple)) else [shape])]


def zeros(shape, dtype=None):
    return T(lambda ev: torch.zeros(_resolve_shape(shape, ev), dtype=F))


def ones(shape, dtype=None):
    return T(lambda ev: torch.ones(_resolve_shape(shape, ev), dtype=F))


def ones_like(x):
    return T(lambda ev: torch.ones_like(_t(ev(x))))


def random_normal(shape):
    return T(lambda ev: torch.as_tensor(_rng.standard_normal(_resolve_shape(shape, ev))), dynamic_batch=False)


def shape(x):
    return T(lambda ev: torch.tensor(list(_t(ev(x)).shape)), dynamic_batch=False)


def reshape(x, shp):
    return T(lambda ev: _t(ev(x)).reshape(_resolve_shape(shp, ev)))


def matmul(a, b):
    a, b = as_T(a), as_T(b)
    return T(lambda ev: _t(ev(a)) @ _t(ev(b)))


def square(x): return T(lambda ev: _t(ev(x)) ** 2)
def sqrt(x): return T(lambda ev: torch.sqrt(_t(ev(x))))
def log(x): return T(lambda ev: torch.log(_t(ev(x))))
def stop_gradient(x): return T(lambda ev: _t(ev(as_T(x))).detach())
def to_float(x): return T(lambda ev: _t(ev(x)).to(F))
def to_int32(x): return T(lambda ev: _t(ev(x)).to(torch.int64))
def expand_dims(x, d): return T(lambda ev: _t(ev(x)).unsqueeze(d))
def range(n): return T(lambda ev: torch.arange(int(n)), dynamic_batch=False)
def argmax(x, axis): return T(lambda ev: torch.argmax(_t(ev(x)), axis))      # ASSUMED: first index on ties
def minimum(a, b): a, b = as_T(a), as_T(b); return T(lambda ev: torch.minimum(_t(ev(a)) + 0 * _t(ev(b)), _t(ev(b)) + 0 * _t(ev(a))))
def no_op(): return T(lambda ev: None, dynamic_batch=False)


def equal(a, b):
    a = as_T(a) if not isinstance(a, T) else a
    if isinstance(b, str):
        return T(lambda ev: ev(a) == b, dynamic_batch=False)
    b = as_T(b)
    return T(lambda ev: _t(ev(a)) == _t(ev(b)))


def _axes(x, axis):
    return None if axis is None else (tuple(axis) if isinstance(axis, (list, tuple)) else (axis,))


def reduce_sum(x, axis=None):
    x = as_T(x)
    return T(lambda ev: _t(ev(x)).sum() if axis is None else _t(ev(x)).sum(_axes(x, axis)))


def reduce_mean(x, axis=None):
    x = as_T(x)
    return T(lambda ev: _t(ev(x)).mean() if axis is None else _t(ev(x)).mean(_axes(x, axis)))


def reduce_max(x, axis=None):
    return T(lambda ev: _t(ev(x)).amax(_axes(x, axis)))


def concat(dim, values):                          # pre-1.0 argument order
    vals = [as_T(v) for v in values]
    return T(lambda ev: torch.cat([_t(ev(v)) for v in vals], dim))


def cond(pred, fn_true, fn_false):
    a, b = fn_true(), fn_false()                   # both branches are graph; only the taken one is evaluated
    return T(lambda ev: ev(a) if bool(ev(pred)) else ev(b), ctrl=[])


def assign(var, value):
    value = as_T(value)

    def fn(ev):
        v = _t(ev(value))
        if _active:                                # (graph construction evaluates with no run active: no effect)
            _active[-1].assigns.append((var, v))
        return v
    return T(fn, dynamic_batch=False)


@contextlib.contextmanager
def control_dependencies(deps):
    _ctrl_stack.append(list(deps))
    try:
        yield
    finally:
        _ctrl_stack.pop()


def group(*ops):
    ops = [o for o in ops if o is not None]
    return T(lambda ev: [ev(o) for o in ops] and None, dynamic_batch=False)


class _NN:
    @staticmethod
    def conv2d(x, k, strides, padding):
        # ASSUMED: cross-correlation, SAME = (k - 1) // 2 zeros before (odd supports), NHWC / HWIO
        assert tuple(strides) == (1, 1, 1, 1) and padding == 'SAME'
        return T(lambda ev: torch.nn.functional.conv2d(_t(ev(x)).permute(0, 3, 1, 2), _t(ev(k)).permute(3, 2, 0, 1),
                                                       padding=((_t(ev(k)).shape[0] - 1) // 2, (_t(ev(k)).shape[1] - 1) // 2)).permute(0, 2, 3, 1))

    @staticmethod
    def max_pool(x, ksize, strides, padding):
        # ASSUMED: 2x2 / 2 on even maps = no padding; gradient to the FIRST maximum of a window
        assert tuple(ksize) == (1, 2, 2, 1) and tuple(strides) == (1, 2, 2, 1)

        def fn(ev):
            v = _t(ev(x))
            n, h, w, c = v.shape
            win = v.reshape(n, h // 2, 2, w // 2, 2, c).permute(0, 1, 3, 5, 2, 4).reshape(n, h // 2, w // 2, c, 4)
            arg = torch.from_numpy(np.argmax(win.detach().numpy(), axis=4))        # first occurrence
            return torch.gather(win, 4, arg[..., None])[..., 0]
        return T(fn)

    @staticmethod
    def relu(x): return T(lambda ev: torch.relu(_t(ev(x))))

    @staticmethod
    def softmax(x): return T(lambda ev: torch.softmax(_t(ev(x)), -1))

    @staticmethod
    def moments(x, axes):
        # ASSUMED: population (biased) variance
        m = T(lambda ev: _t(ev(x)).mean(tuple(axes)))
        v = T(lambda ev: ((_t(ev(x)) - _t(ev(m))) ** 2).mean(tuple(axes)))
        return m, v


nn = _NN()


class _Image:
    @staticmethod
    def resize_images(x, size):
        # ASSUMED: legacy bilinear, align_corners=False, integer ratio = strided pick of pixel (r*i, r*j)
        def fn(ev):
            v = _t(ev(x))
            r = v.shape[1] // size[0]
            assert v.shape[1] == size[0] * r and v.shape[2] == size[1] * r
            return v[:, ::r, ::r, :]
        return T(fn)


image = _Image()


class _Momentum:
    """ASSUMED: accum = momentum * accum + grad; var -= lr * accum (non-Nesterov)."""

    def __init__(self, learning_rate, momentum):
        self.lr, self.mu = as_T(learning_rate), as_T(momentum)
        self.accum = {}

    def compute_gradients(self, cost):
        vs = [v for v in _all_variables if v.trainable]
        bundle = T(lambda ev: torch.autograd.grad(_t(ev(cost)), [_t(ev(v)) for v in vs], allow_unused=True, retain_graph=True),
                   dynamic_batch=False)
        used = bundle.value
        out = []
        for k, v in enumerate(vs):
            if used[k] is None:
                out.append((None, v))
            else:
                out.append((T((lambda kk: lambda ev: ev(bundle)[kk])(k), dynamic_batch=False), v))
        return out

    def apply_gradients(self, grads_and_vars):
        gv = [(as_T(g), v) for g, v in grads_and_vars]

        def fn(ev):
            if not _active:
                return None
            lr, mu = float(_t(ev(self.lr))), float(_t(ev(self.mu)))
            for g, v in gv:
                gval = _t(ev(g)).detach()
                acc = mu * self.accum.get(id(v), torch.zeros_like(gval)) + gval
                self.accum[id(v)] = acc
                _active[-1].assigns.append((v, _t(ev(v)).detach() - lr * acc))
            return None
        return T(fn, dynamic_batch=False)

    def minimize(self, cost):
        return self.apply_gradients([(g, v) for g, v in self.compute_gradients(cost) if g is not None])


class _Train:
    MomentumOptimizer = _Momentum


train = _Train()


def reset():
    del _all_variables[:]
    global _rng
    _rng = np.random.RandomState(0)
