"""Whole-net CPU restatement of the reference's TensorFlow graph.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``); PARITY UNPINNED (TensorFlow
cannot run here).  Walks a linked product ``Net`` (the Layer tree is plain
data: class names, hypers, tree links) and re-evaluates it the way the
reference builds its graph, with torch-CPU tensors and torch autograd standing
in for TensorFlow ops and ``tf.gradients``.  Each operator below is
cross-checked against the hand-written float64 NumPy forms in ``np_ops``
(tests/test_oracle_vs_torch.py), known answers KA1..KA6 of SURVEY.md 8c are
checked in tests/test_oracle_known_answers.py, and tests/test_ref_graph_golden.py
holds it to vectors that the reference's OWN graph-assembly code produced over a
stand-in for its TensorFlow calls (that pins the Python on top of the operators,
not TensorFlow's operator semantics: hence still "unpinned").

Reference lines followed (paths relative to /root/reference/scripts):
  lib/layer_types.py: LinTrans :39-53, Rect :76-79, Softmax :81-84,
    ToPyramid :118-125, MultiscaleConvMax :149-194, MultiscaleRect :196-199,
    Select :201-206, BatchNorm :219-239, MultiscaleBatchNorm :241-249,
    CrossEntropyError :262-272, Chain :299-310
  lib/net_types.py: minimize_expectation :24-37, Net.link :56-63, SRNet :85-97,
    ActorNet :103-181, CriticNet :187-284
"""
from functools import reduce

import numpy as np
import torch
import torch.nn.functional as TF


def _nchw(t):
    return t.permute(0, 3, 1, 2)


def _nhwc(t):
    return t.permute(0, 2, 3, 1)


def conv_same(x, w):                      # tf.nn.conv2d(x, w, (1,1,1,1), 'SAME'), 3x3
    return _nhwc(TF.conv2d(_nchw(x), w.permute(3, 2, 0, 1), padding=(w.shape[0] - 1) // 2))


def pool2(x):                             # tf.nn.max_pool 2x2/2 'SAME' on even maps
    return _nhwc(TF.max_pool2d(_nchw(x), 2))


def pool2_forced(x, arg):
    """2x2/2 max-pool with the window element GIVEN (arg in 0..3 = 2*dy + dx, [n, H/2, W/2, C]):
    value and gradient go through that element.  Decision-forced comparison (see RefNet.forward)."""
    n, H, W, C = x.shape
    win = x.reshape(n, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(n, H // 2, W // 2, C, 4)
    return torch.gather(win, 4, torch.as_tensor(arg, dtype=torch.int64)[..., None])[..., 0]


class RefNet:
    """Evaluates loss, statistics and the TALR-momentum update of a product Net."""

    def __init__(self, net, dtype=torch.float64):
        self.net = net
        self.dtype = dtype
        self.kind = net._net_kind
        self.θ = {}          # id(Param) -> leaf tensor
        self.accum = {}
        self.state = {}      # id(Param) -> tensor for m_avg / v_avg
        self.forced = {}
        self._at = (None, None)

    # ---- parameters -----------------------------------------------------------
    def load_params(self, values=None):
        """values: dict id(Param) -> ndarray; default: read the product net's device buffers."""
        for p in self.net._all_params:
            v = values[id(p)] if values is not None else p.numpy()
            t = torch.tensor(np.asarray(v, np.float64).reshape(p.shape), dtype=self.dtype)
            if p.trainable:
                self.θ[id(p)] = t.requires_grad_(True)
                self.accum[id(p)] = torch.zeros_like(t)
            else:
                self.state[id(p)] = t

    def V(self, p):
        return self.θ[id(p)] if p.trainable else self.state[id(p)]

    # ---- layer semantics ----------------------------------------------------------
    def _link(self, ℓ, x, y, mode, out):
        """Returns the layer's x and fills out[ℓ] = dict(c_err, c_mod, n_ops, δ_cor)."""
        name = type(ℓ).__name__
        rec = dict(c_err=0.0, c_mod=0.0, n_ops=0)
        θ, ϕ = ℓ.params, ℓ.hypers
        forced = self.forced
        if name == 'Chain':
            for k, c in enumerate(ℓ.comps):
                self._at = (id(ℓ), k)            # position of the component being linked (keys of `forced`)
                x = self._link(c, x, y, mode, out)
            rec['c_err'] = sum(out[id(c)]['c_err'] for c in ℓ.comps)
            rec['c_mod'] = sum(out[id(c)]['c_mod'] for c in ℓ.comps)
            rec['n_ops'] = sum(out[id(c)]['n_ops'] for c in ℓ.comps)
            if ℓ.comps and 'δ_cor' in out[id(ℓ.comps[-1])]:
                rec['δ_cor'] = out[id(ℓ.comps[-1])]['δ_cor']
        elif name == 'ToPyramid':
            x = [x[:, ::2 ** i, ::2 ** i, :] for i in range(ϕ.n_scales)]
        elif name == 'MultiscaleConvMax':
            L = len(ϕ.n_chan)
            wh = [self.V(getattr(θ, 'w_horz_%i' % i)) for i in range(L)]
            wv = [self.V(getattr(θ, 'w_vert_%i' % i)) for i in range(L - 1)]
            b = [self.V(getattr(θ, 'b_%i' % i)) for i in range(L)]
            xs = list(x)[-L:]
            o = [b[0] + conv_same(xs[0], wh[0])]
            for i in range(1, L):
                arg = forced.get(('pool', id(ℓ), i - 1)) if forced else None
                pooled = pool2(o[i - 1]) if arg is None else pool2_forced(o[i - 1], arg)
                o.append(b[i] + conv_same(xs[i], wh[i]) + conv_same(pooled, wv[i - 1]))
            rec['c_mod'] = ϕ.k_l2 * (sum((w ** 2).sum() for w in wh) + sum((w ** 2).sum() for w in wv))
            rec['n_ops'] = sum(o[i].shape[1] * o[i].shape[2] * (wh[i].numel() + (wv[i - 1].numel() if i > 0 else 0))
                               for i in range(L))
            rec['pre_bn'] = o
            x = o
        elif name == 'Conv':                                   # layer_types.py:55-74
            w, b = self.V(θ.w), self.V(θ.b)
            x = conv_same(x, w) + b
            w_eq = torch.tensor(θ.w.eq, dtype=self.dtype) if θ.w.eq is not None else 0
            rec['c_mod'] = ϕ.k_l2 * ((w - w_eq) ** 2).sum()
            rec['n_ops'] = x.shape[1] * x.shape[2] * w.numel()
        elif name == 'MultiscaleBatchNorm':
            x = [self._link(c, x_i, y, mode, out) for c, x_i in zip(ℓ.comps, x)]
        elif name == 'BatchNorm':
            γ, β = self.V(θ.γ), self.V(θ.β)
            dims = tuple(range(x.dim() - 1))
            if mode == 'tr':
                m = x.mean(dims)
                v = ((x - m) ** 2).mean(dims)                       # tf.nn.moments: biased
                rec['new_avg'] = (ϕ.d * self.state[id(θ.m_avg)] + (1 - ϕ.d) * m.detach(),
                                  ϕ.d * self.state[id(θ.v_avg)] + (1 - ϕ.d) * v.detach())
                x = γ * (x - m) / torch.sqrt(v + ϕ.ϵ) + β
            else:
                x = γ * (x - self.state[id(θ.m_avg)]) / torch.sqrt(self.state[id(θ.v_avg)] + ϕ.ϵ) + β
        elif name == 'MultiscaleRect':
            masks = forced.get(('relu',) + self._at) if forced else None
            x = [torch.relu(x_i) if masks is None else x_i * torch.as_tensor(m, dtype=self.dtype)
                 for x_i, m in zip(x, masks if masks is not None else x)]
        elif name == 'Rect':
            m = forced.get(('relu',) + self._at) if forced else None
            x = torch.relu(x) if m is None else x * torch.as_tensor(m, dtype=self.dtype)
        elif name == 'MaxPool':                                # layer_types.py:86-94
            # tf.nn.max_pool(x, strides, k_shape, 'SAME'): the reference hands its hypers over in this order, so
            # TensorFlow's ksize is hypers.stride and its strides are hypers.supp.  SAME: out = ceil(H / step),
            # pad_before = pad_total // 2, padded cells never win; gradient to the first maximum of a window.
            win, step = int(ϕ.stride), int(ϕ.supp)
            H, W = x.shape[1], x.shape[2]
            ho, wo = -(-H // step), -(-W // step)
            ty, tx = max((ho - 1) * step + win - H, 0), max((wo - 1) * step + win - W, 0)
            xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (tx // 2, tx - tx // 2, ty // 2, ty - ty // 2), value=float('-inf'))
            x = torch.nn.functional.max_pool2d(xp, win, step).permute(0, 2, 3, 1)
        elif name == 'GlobalMaxPool':                          # layer_types.py:96-100: tf.reduce_max over H, W
            x = x.amax(dim=(1, 2))                             # (ties share the gradient, as in TensorFlow)
        elif name == 'Select':
            x = x[ϕ.i]
        elif name == 'LinTrans':
            w, b = self.V(θ.w), self.V(θ.b)
            x = x.reshape(x.shape[0], -1) @ w + b
            w_eq = torch.tensor(θ.w.eq, dtype=self.dtype) if θ.w.eq is not None else 0
            rec['c_mod'] = ϕ.k_l2 * ((w - w_eq) ** 2).sum()
            rec['n_ops'] = w.shape[0] * w.shape[1]
        elif name == 'Softmax':
            x = torch.softmax(x, 1)
        elif name == 'CrossEntropyError':
            n_cls = y.shape[1]
            p_cls = ϕ.ϵ / n_cls + (1 - ϕ.ϵ) * x
            rec['c_err'] = -(y * torch.log(p_cls)).sum(1)
            rec['δ_cor'] = (torch.argmax(x, 1) == torch.argmax(y, 1)).to(self.dtype)
        elif name == 'NoOp':
            pass
        else:
            raise NotImplementedError(name)
        rec['x'] = x
        out[id(ℓ)] = rec
        return x

    # ---- whole graph -------------------------------------------------------------------
    def forward(self, x0, y, mode='ev', τ=None, ϵ=None, k_cpt=None, forced=None):
        """forced: the DISCRETE decisions of another evaluation of the same graph (the product's):
        {('pool', id(MultiscaleConvMax), i): window arg-max of pre-BN map i,
         ('relu', id(Chain), k): 0/1 mask (list of masks for MultiscaleRect) of the chain's k-th component}.
        Max-pool and ReLU then take the given branch (value and gradient) instead of deciding
        themselves: fp32 and float64 disagree on a near-tie now and then, and ONE flipped element
        moves some gradient tensors by tens of percent -- with the decisions pinned every tensor must
        agree to rounding, with no outlier band.  Keys that are absent decide freely."""
        self.forced = forced or {}
        net, ϕ = self.net, self.net.hypers
        T = lambda a: torch.as_tensor(np.asarray(a, np.float64), dtype=self.dtype)
        x0, y = T(x0), T(y)
        n = x0.shape[0]
        out = {}
        dyn = bool(getattr(ϕ, 'dyn_k_cpt', False))
        if self.kind != 'sr':
            τ = ϕ.τ if τ is None else τ
            ϵ = ϕ.ϵ if ϵ is None else ϵ
            k_cpt = (T(k_cpt) * torch.ones(n, dtype=self.dtype)) if dyn else ϕ.k_cpt

        def link_layer(ℓ, x):
            x_l = self._link(ℓ, x, y, mode, out)
            if ℓ.router is not None:
                cat = lambda x_: torch.cat([x_.reshape(n, -1), (ϕ.α_cpt * k_cpt)[:, None]], 1)
                x_rte = x_l if not dyn else (list(map(cat, x_l)) if isinstance(x_l, list) else cat(x_l))
                self._link(ℓ.router, x_rte, y, mode, out)
            for s in ℓ.sinks:
                link_layer(s, x_l)
        link_layer(net.root, x0)
        R = lambda ℓ: out[id(ℓ)]
        layers = list(net.layers)
        ones = torch.ones(n, dtype=self.dtype)
        n_leaves = lambda ℓ: 1 if not ℓ.sinks else sum(map(n_leaves, ℓ.sinks))
        res = dict(out=out, layers=layers)

        if self.kind == 'sr':
            for ℓ in layers:
                R(ℓ)['p_ev'] = ones
                R(ℓ)['p_tr'] = ones
            res['c_tot'] = (sum(R(ℓ)['c_err'] + R(ℓ)['c_mod'] for ℓ in layers) * ones).mean()
            return res

        root_leaves = n_leaves(net.root)
        p_ϵ = lambda ℓ: ϵ * n_leaves(ℓ) / root_leaves
        critic = self.kind == 'critic'

        def route(ℓ, p_tr, p_ev):
            r = R(ℓ)
            r['p_tr'], r['p_ev'] = p_tr, p_ev
            c_err = ((1 - r.get('δ_cor', 1)) if getattr(ϕ, 'use_cls_err', False) else r['c_err']) if critic else None
            if len(ℓ.sinks) < 2:
                for s in ℓ.sinks:
                    route(s, p_tr, p_ev)
                if critic:
                    r['c_ev'] = c_err + k_cpt * r['n_ops'] + sum(R(s)['c_ev'] for s in ℓ.sinks)
                    r['c_opt'] = c_err + k_cpt * r['n_ops'] + sum(R(s)['c_opt'] for s in ℓ.sinks)
                    r['c_cre'] = 0.0
                return
            rx = R(ℓ.router)['x']
            π_tr = ((1 - p_ϵ(ℓ) / p_tr[:, None]) * torch.softmax(rx / τ, 1)
                    + torch.tensor([p_ϵ(s) for s in ℓ.sinks], dtype=self.dtype) / p_tr[:, None])
            π_ev = (torch.argmax(rx, 1)[:, None] == torch.arange(len(ℓ.sinks))).to(self.dtype)
            for i, s in enumerate(ℓ.sinks):
                route(s, p_tr * π_tr[:, i], p_ev * π_ev[:, i])
            if critic:
                rops = R(ℓ.router)['n_ops']
                r['c_ev'] = c_err + k_cpt * (r['n_ops'] + rops) + sum(π_ev[:, i] * R(s)['c_ev'] for i, s in enumerate(ℓ.sinks))
                r['c_opt'] = c_err + k_cpt * (r['n_ops'] + rops) + reduce(torch.minimum, (R(s)['c_opt'] * ones for s in ℓ.sinks))
                r['c_cre'] = ϕ.k_cre * sum(
                    (rx[:, i] + (R(s)['c_opt'] if ϕ.optimistic else R(s)['c_ev']).detach()) ** 2
                    for i, s in enumerate(ℓ.sinks))
        route(net.root, ones, ones)
        rops = lambda ℓ: R(ℓ.router)['n_ops'] if ℓ.router is not None else 0
        rmod = lambda ℓ: R(ℓ.router)['c_mod'] if ℓ.router is not None else 0
        switches = [ℓ for ℓ in layers if len(ℓ.sinks) > 1]
        if not critic:
            c_err = sum(R(ℓ)['p_tr'] * R(ℓ)['c_err'] for ℓ in layers)
            c_cpt = sum(R(ℓ)['p_tr'] * k_cpt * (R(ℓ)['n_ops'] + rops(ℓ)) for ℓ in layers)
            c_mod = sum(R(ℓ)['p_tr'].detach() * (R(ℓ)['c_mod'] + rmod(ℓ)) for ℓ in layers)
            c_dec = sum(R(ℓ)['p_tr'].detach() * ϕ.k_dec * (R(ℓ.router)['x'] ** 2).sum(1) for ℓ in switches)
            res['c_tot'] = (c_err + c_cpt + c_mod + c_dec).mean()
        else:
            c_err = sum(R(ℓ)['p_tr'].detach() * R(ℓ)['c_err'] for ℓ in layers)
            c_cre = sum(R(ℓ)['p_tr'].detach() * R(ℓ)['c_cre'] for ℓ in layers)
            c_mod = sum(R(ℓ)['p_tr'].detach() * (R(ℓ)['c_mod'] + rmod(ℓ)) for ℓ in layers)
            res['c_tot'] = (c_err + c_cre + c_mod).mean()
        return res

    def stats(self, res):
        """acc / moc / per-leaf statistics (scripts/train-nets:117-130), per sample."""
        R = lambda ℓ: res['out'][id(ℓ)]
        layers = res['layers']
        leaves = [ℓ for ℓ in layers if not ℓ.sinks]
        rops = lambda ℓ: R(ℓ.router)['n_ops'] if ℓ.router is not None else 0
        return dict(
            acc=sum(R(ℓ)['p_ev'] * R(ℓ)['δ_cor'] for ℓ in leaves).detach().numpy(),
            moc=sum(R(ℓ)['p_ev'] * float(R(ℓ)['n_ops'] + rops(ℓ)) for ℓ in layers).detach().numpy(),
            p_leaf=np.stack([(R(ℓ)['p_ev']).detach().numpy() for ℓ in leaves]))

    # ---- training step -------------------------------------------------------------------
    def train_step(self, x0, y, λ_lrn, μ_lrn=None, τ=None, k_cpt=None, forced=None):
        """One net.train.run: gradients, TALR scaling, momentum update, BN moving
        averages.  Returns the forward result dict (with 'grads' added)."""
        from lib.net_types import params_list_rec
        net, ϕ = self.net, self.net.hypers
        μ = ϕ.μ_lrn if μ_lrn is None else μ_lrn
        for t in self.θ.values():
            t.grad = None
        res = self.forward(x0, y, 'tr', τ=τ, k_cpt=k_cpt, forced=forced)
        bns = [(k, r['x']) for k, r in res['out'].items() if 'new_avg' in r and getattr(r['x'], 'requires_grad', False)]
        reach = torch.autograd.grad(res['c_tot'], [x for _, x in bns], allow_unused=True, retain_graph=True) if bns else []
        used_bn = {k for (k, _), g in zip(bns, reach) if g is not None}       # BatchNorms whose output reaches the loss
        res['c_tot'].backward()
        R = lambda ℓ: res['out'][id(ℓ)]
        scale = {}
        talr = self.kind != 'sr' and getattr(ϕ, 'talr', False)
        for ℓ in res['layers']:
            s = float(1 / torch.sqrt((R(ℓ)['p_tr'].detach() ** 2).mean())) if talr else 1.0
            for p in params_list_rec(ℓ):
                scale[id(p)] = s
            for p in params_list_rec(ℓ.router):
                scale[id(p)] = ϕ.α_rtr * s      # (α_rtr * lr_scale, lr_scale = 1 without TALR: net_types.py:25-33)
        res['grads'] = {}
        with torch.no_grad():
            for p in net._all_params:
                if not p.trainable:
                    continue
                t = self.θ[id(p)]
                g = t.grad if t.grad is not None else torch.zeros_like(t)
                res['grads'][id(p)] = g.clone()
                self.accum[id(p)] = μ * self.accum[id(p)] + scale[id(p)] * g
                t -= λ_lrn * self.accum[id(p)]
            for rec_id, rec in res['out'].items():
                pass
            # BN moving averages.  TensorFlow only executes what the fetched op needs: the two tf.assign
            # of a BatchNorm hang off its OUTPUT by control dependency (layer_types.py:233-236), so a
            # BatchNorm whose output feeds nothing that reaches the loss -- a scale the child block drops
            # (negative indexing, :163) and no exit reads (Select(-1)) -- never moves its averages.
            # (Found by running the reference's own graph code: tests/test_ref_graph_golden.py.)
            def upd(ℓ):
                if ℓ is None:
                    return
                if type(ℓ).__name__ == 'BatchNorm' and 'new_avg' in res['out'].get(id(ℓ), {}) and id(ℓ) in used_bn:
                    m, v = res['out'][id(ℓ)]['new_avg']
                    self.state[id(ℓ.params.m_avg)], self.state[id(ℓ.params.v_avg)] = m, v
                for c in ℓ.comps:
                    upd(c)
            for ℓ in res['layers']:
                upd(ℓ)
                upd(ℓ.router)
        return res
