"""Net types: statically-routed, actor and critic networks, MI355X-native.

Mirrors ``scripts/lib/net_types.py`` of the reference: same class names,
constructor keywords, placeholders (``x0, y, mode, λ_lrn, μ_lrn, ϵ, τ, k_cpt``),
``train`` op and tree iterators, so ``arch_and_hypers.py`` and a ``train-nets``
style driver work unchanged:

    net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: lr, net.τ: τ})

What differs: the reference assembles a TensorFlow graph (routing products,
expected costs, ``minimize_expectation``) and lets TF differentiate it; here
``link`` only infers shapes and registers parameters, and the arithmetic is a
static plan of hand-written HIP kernel launches (``lib/_plan.py``):

* routing probabilities, costs and their gradients -> ``mpnn_route``
  (reference: net_types.py:108-131, 165-177, 193-243, 273-280);
* TALR scaling + L2 + momentum -> ``mpnn_talr_momentum_step``
  (reference: net_types.py:24-37 and tf.train.MomentumOptimizer).
"""
from abc import ABCMeta
from types import SimpleNamespace as Ns

import numpy as np

from lib.layer_types import BatchNorm, Chain, Layer, LinTrans, NoOp, Rect, Sym, _Linker

################################################################################
# Support Functions  (reference: net_types.py:14-22)
################################################################################

def n_leaves(ℓ):
    return 1 if len(ℓ.sinks) == 0 else sum(map(n_leaves, ℓ.sinks))


def params_list_rec(ℓ):
    if ℓ is not None:
        yield from vars(ℓ.params).values()
        for c in getattr(ℓ, 'comps', []):
            yield from params_list_rec(c)

################################################################################
# Placeholders and the train op (session-free stand-ins for the TF idioms)
################################################################################

class Placeholder:
    """A feedable input; hashable so it can key a feed dict like a tf.placeholder."""

    def __init__(self, name, default=None):
        self.name = name
        self.default = default

    def __repr__(self):
        return '<placeholder %s>' % self.name


class _TrainOp:
    def __init__(self, net):
        self._net = net

    def run(self, feed):
        return self._net._run_train(feed)

    def run_steps(self, feeds):
        """len(feeds) consecutive training steps, replayed as ONE hipGraph where the engine can (Engine.run_steps);
        the same results as calling run() on each feed in turn."""
        eng = self._net.engine()
        if hasattr(eng, 'run_steps'):
            return eng.run_steps(list(feeds))
        for f in feeds:                                   # (the single-scale Conv engine: step by step)
            eng.run(f, train=True)

################################################################################
# Root Network Class  (reference: net_types.py:43-79)
################################################################################

class Net(metaclass=ABCMeta):
    default_hypers = Ns(x0_shape=(), y_shape=())
    _net_kind = 'sr'

    def __init__(self, **options):
        self.root = options.pop('root', NoOp())
        self.hypers = Ns(**{**vars(type(self).default_hypers), **options})
        self.params = Ns()
        self.x0 = Placeholder('x0')
        self.y = Placeholder('y')
        self.mode = Placeholder('mode', 'ev')
        self.train = _TrainOp(self)
        self._engine = None
        self._device = None
        self._dp = None
        with _Linker() as lk:
            self.link()
        self._all_params = lk.params

    # -- graph construction ----------------------------------------------------
    def _k_cpt_dyn(self):
        return bool(getattr(self.hypers, 'dyn_k_cpt', False))

    def link(self):
        dyn = self._k_cpt_dyn()

        def with_k_cpt(x_):                       # reference: net_types.py:149-154
            s = Sym((x_.n_el + 1,), x_.producer)
            s.base = x_
            return s

        def link_layer(ℓ, x, y, mode):
            ℓ.link(x, y, mode)
            if ℓ.router is not None:
                if not dyn:
                    x_rte = ℓ.x
                elif isinstance(ℓ.x, list):
                    x_rte = list(map(with_k_cpt, ℓ.x))
                else:
                    x_rte = with_k_cpt(ℓ.x)
                ℓ.router.dyn_k_cpt = dyn
                ℓ.router.link(x_rte, y, mode)
            for s in ℓ.sinks:
                link_layer(s, ℓ.x, y, mode)
        link_layer(self.root, Sym(self.hypers.x0_shape), Sym(self.hypers.y_shape), self.mode)

    @property
    def layers(self):
        def all_in_tree(layer):
            yield layer
            for sink in layer.sinks:
                yield from all_in_tree(sink)
        yield from all_in_tree(self.root)

    @property
    def leaves(self):
        return (ℓ for ℓ in self.layers if len(ℓ.sinks) == 0)

    @property
    def switches(self):
        return (ℓ for ℓ in self.layers if len(ℓ.sinks) > 1)

    # -- execution ---------------------------------------------------------------
    def to(self, device):
        """Pin the net to a device before first use (default: cuda:LOCAL_RANK)."""
        self._device = device
        return self

    def engine(self):
        if self._engine is None:
            from lib._plan import Engine          # imports the HIP library; raises if missing
            from lib._plan_conv import ConvEngine, is_conv_net
            # (a statically-routed chain of single-scale Conv layers has its own, simpler plan)
            self._engine = (ConvEngine if is_conv_net(self) else Engine)(self, self._device)
        return self._engine

    def _run_train(self, feed):
        return self.engine().run(feed, train=True)

    def eval(self, feed, routed=False):
        """Forward pass + routing in evaluation mode ('ev': BatchNorm moving averages, hard routing);
        per-layer results are then readable as ``ℓ.p_ev``, ``ℓ.δ_cor`` ... device tensors.

        routed='auto' picks the routed evaluation from 512 samples per launch on and the dense one below
        (Engine.routed_min_batch); routed=True runs the ROUTED evaluation: a block only processes the samples its ancestors'
        routers sent to it (sample lists compacted on the device, no host sync) -- from a depth the engine picks by
        batch size (Engine.routed_prefix: the first blocks lose few samples, and below ~6 000 samples running their
        convs on everybody in wavefront-grouped launches is faster than gathering); routed=<int d> sets that depth
        (1: every block below the root gathers).  ``p_ev`` and every
        p_ev-weighted statistic (acc, moc, p_cor, p_inc, *_by_cls: all that the reference's figure
        scripts read) are identical to the dense pass; per-leaf ``c_err`` / ``δ_cor`` and ``router.x``
        are those of the dense pass where the sample reaches the node and 0 elsewhere."""
        return self.engine().run(feed, train=False, routed=routed)

    def state(self):
        """Per-sample statistics of the last run, keyed like the reference's
        ``state_tensors`` (scripts/train-nets:117-130)."""
        return self.engine().state()

################################################################################
# Statically-Routed Networks  (reference: net_types.py:85-97)
################################################################################

class SRNet(Net):
    default_hypers = Ns(λ_lrn=1e-3, μ_lrn=0.9, seed=None)
    _net_kind = 'sr'

    def link(self):
        super().link()
        ϕ = self.hypers
        self.λ_lrn = Placeholder('λ_lrn', ϕ.λ_lrn)
        self.μ_lrn = Placeholder('μ_lrn', ϕ.μ_lrn)

################################################################################
# Actor Networks  (reference: net_types.py:103-181)
################################################################################

class ActorNet(Net):
    default_hypers = Ns(
        k_cpt=0.0, k_dec=0.01, ϵ=1e-6, τ=1.0, λ_lrn=1e-3, μ_lrn=0.9,
        dyn_k_cpt=False, α_cpt=1e7, talr=True, α_rtr=1.0, seed=None)
    _net_kind = 'actor'

    def link(self):
        ϕ = self.hypers
        self.λ_lrn = Placeholder('λ_lrn', ϕ.λ_lrn)
        self.μ_lrn = Placeholder('μ_lrn', ϕ.μ_lrn)
        self.ϵ = Placeholder('ϵ', ϕ.ϵ)
        self.τ = Placeholder('τ', ϕ.τ)
        self.k_cpt = Placeholder('k_cpt') if ϕ.dyn_k_cpt else ϕ.k_cpt
        super().link()

################################################################################
# Critic Networks  (reference: net_types.py:187-284)
################################################################################

class CriticNet(Net):
    default_hypers = Ns(
        k_cpt=0.0, k_cre=1e-3, ϵ=1e-6, τ=0.01, optimistic=False,
        dyn_k_cpt=False, α_cpt=1e7, use_cls_err=False, λ_lrn=1e-3, μ_lrn=0.9,
        talr=True, α_rtr=1.0, seed=None)
    _net_kind = 'critic'

    def link(self):
        ϕ = self.hypers
        self.λ_lrn = Placeholder('λ_lrn', ϕ.λ_lrn)
        self.μ_lrn = Placeholder('μ_lrn', ϕ.μ_lrn)
        self.ϵ = Placeholder('ϵ', ϕ.ϵ)
        self.τ = Placeholder('τ', ϕ.τ)
        self.k_cpt = Placeholder('k_cpt') if ϕ.dyn_k_cpt else ϕ.k_cpt
        super().link()
