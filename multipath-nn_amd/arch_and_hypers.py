"""Architecture, hyper-parameters and net constructors of the shipped experiments.

Same public names as the reference's ``scripts/arch_and_hypers.py`` (``arch``,
``k_cpts``, ``batch_size``, ``n_iter``, ``t_log``, ``λ_lrn``, ``τ_cr``, ``τ_ds``,
``router``, ``pyr``, ``rcm``, ``reg``, ``sr_chain``, ``ac_chain``, ``cr_chain``,
``ac_tree``, ``cr_tree``); the reference's own file also runs unchanged on top of
``lib/`` (tests/test_host_cpu.py::test_reference_spec_file_drops_in_unchanged).  Values: reference
arch_and_hypers.py:12-39; builders :45-70; constructors :76-139.
"""
from lib.layer_types import (
    BatchNorm, Chain, CrossEntropyError, LinTrans, MultiscaleBatchNorm,
    MultiscaleConvMax, MultiscaleLLN, MultiscaleRect, Rect, Select, Softmax,
    ToPyramid)
from lib.net_types import ActorNet, CriticNet, SRNet

# ---- network hyper-parameters ------------------------------------------------
conv_supp = 3
router_n_chan = 16
k_cpts = [0.0] + [1e-9 * 2 ** i for i in range(7)]       # 0, 1e-9 ... 6.4e-8
k_l2 = 1e-4
σ_w = 1
arch = [4 * [16], 4 * [16], 3 * [32], 3 * [32], 2 * [64], 2 * [64], [128], [128]]

# ---- training hyper-parameters -------------------------------------------------
n_iter = 80000
t_log = 2500
batch_size = 128


def λ_lrn(t):
    return 0.1 / 2 ** (t / 10000)


def τ_cr(t):
    return 0.1 / 2 ** (t / 20000)


def τ_ds(t):
    return 1 / 2 ** (t / 20000)

# ---- components ---------------------------------------------------------------------

def _dense(n_out, **kw):
    return LinTrans(n_chan=n_out, k_l2=k_l2, **{'σ_w': σ_w, **kw})


def router(n_sinks):
    """Coarsest scale -> 16 -> 16 -> n_sinks MLP; the last map starts at zero."""
    if n_sinks < 2:
        return None
    hidden = []
    for _ in range(2):
        hidden += [_dense(router_n_chan), BatchNorm(), Rect()]
    return Chain(name='Router', comps=[Select(i=-1), *hidden, _dense(n_sinks, σ_w=0)])


def pyr(*sinks):
    return Chain(name='ToPyramid', sinks=sinks, router=router(len(sinks)),
                 comps=[ToPyramid(n_scales=len(arch[0]))])


def rcm(i, *sinks):
    body = [MultiscaleConvMax(n_chan=arch[i], supp=conv_supp, k_l2=k_l2, σ_w=σ_w),
            MultiscaleBatchNorm(), MultiscaleRect()]
    return Chain(name='ReConvMax', sinks=sinks, router=router(len(sinks)), comps=body)


def reg(n_chan):
    return Chain(name='LogReg', comps=[Select(i=-1), _dense(n_chan), Softmax(), CrossEntropyError()])

# ---- constructors -----------------------------------------------------------------------

def sr_chain(n_tf):
    """pyr -> rcm0 -> ... -> rcm(n_tf-1) -> reg."""
    def make_net(x0_shape, y_shape):
        node = reg(y_shape[0])
        for i in range(n_tf - 1, -1, -1):
            node = rcm(i, node)
        return SRNet(x0_shape=x0_shape, y_shape=y_shape, root=pyr(node))
    return make_net


def dr_chain(type_, **hypers):
    """Every block gets an exit classifier (sink 0) and the next block (sink 1)."""
    def make_net(x0_shape, y_shape):
        node = rcm(len(arch) - 1, reg(y_shape[0]))
        for i in range(len(arch) - 2, -1, -1):
            node = rcm(i, reg(y_shape[0]), node)
        return type_(x0_shape=x0_shape, y_shape=y_shape, root=pyr(node), **hypers)
    return make_net


def dr_tree(type_, **hypers):
    """Binary tree over blocks 0-2, then chains 3-7 (47 blocks, 47 leaves)."""
    def make_net(x0_shape, y_shape):
        n_cls = y_shape[0]

        def tail(i=3):
            return rcm(i, reg(n_cls)) if i == len(arch) - 1 else rcm(i, reg(n_cls), tail(i + 1))

        def fork(i):
            kids = (tail(), tail()) if i == 2 else (fork(i + 1), fork(i + 1))
            return rcm(i, reg(n_cls), *kids)
        return type_(x0_shape=x0_shape, y_shape=y_shape, root=pyr(fork(0)), **hypers)
    return make_net


def ac_chain(**hypers):
    return dr_chain(ActorNet, **hypers)


def ac_tree(**hypers):
    return dr_tree(ActorNet, **hypers)


def cr_chain(**hypers):
    return dr_chain(CriticNet, **hypers)


def cr_tree(**hypers):
    return dr_tree(CriticNet, **hypers)
