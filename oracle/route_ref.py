"""Routing, costs and their gradients on a bare routing tree (float64 torch, autograd).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``); PARITY UNPINNED (TensorFlow cannot run here).
The kernel-level oracle of ``mpnn_route``: the same formulas as the whole-net restatement
``ref_net.RefNet.forward`` but on a tree given as plain tables, so 2-, 3- and 4-way switches of
any shape can be checked without building (and convolving through) a net.
``tests/test_oracle_known_answers.py`` checks it against ``RefNet`` on the shipped chains.

Reference lines followed (scripts/lib/net_types.py):
  _route / _route_sinks_stat / _route_sinks_dyn    :108-131 (actor), :193-243 (critic)
  ActorNet cost assembly :165-177; CriticNet :273-280; SRNet :93-95
  minimize_expectation's lr scales :24-27 (returned as node statistics)
"""
from functools import reduce

import numpy as np
import torch


class Tree:
    """nodes: list of dicts(parent, sinks=[child ids], leaf_id or None, switch_id or None) in DFS
    preorder (node 0 = root)."""

    def __init__(self, nodes):
        self.nodes = nodes
        for i, nd in enumerate(nodes):
            nd.setdefault('sinks', [])
        self.leaves = [i for i, nd in enumerate(nodes) if not nd['sinks']]
        self.switches = [i for i, nd in enumerate(nodes) if len(nd['sinks']) > 1]
        for k, i in enumerate(self.leaves):
            nodes[i]['leaf_id'] = k
        for k, i in enumerate(self.switches):
            nodes[i]['switch_id'] = k

    def n_leaves(self, i):
        s = self.nodes[i]['sinks']
        return 1 if not s else sum(self.n_leaves(c) for c in s)

    def tables(self, max_sinks):
        """(node table [n_nodes][8], switch-children table) as mpnn_route wants them."""
        par = {0: (-1, 0)}
        for i, nd in enumerate(self.nodes):
            for k, c in enumerate(nd['sinks']):
                par[c] = (i, k)
        tab, depth = [], {}
        for i in range(len(self.nodes)):
            depth[i] = 0 if par[i][0] < 0 else depth[par[i][0]] + 1
        rank = {j: k for k, j in enumerate(sorted(depth, key=lambda j: (depth[j], j)))}
        for i, nd in enumerate(self.nodes):
            tab += [par[i][0], par[i][1], len(nd['sinks']), nd.get('switch_id', -1) if len(nd['sinks']) > 1 else -1,
                    nd.get('leaf_id', -1) if not nd['sinks'] else -1, self.n_leaves(i), depth[i], rank[i]]
        kids = []
        for i in self.switches:
            row = list(self.nodes[i]['sinks'])
            kids += row + [0] * (max_sinks - len(row))
        return np.array(tab, np.int32), np.array(kids if kids else [0], np.int32)


def route(kind, tree, r, c_err, d_cor, ops, τ=1.0, ϵ=1e-6, k_cpt=0.0, k_dec=0.01, k_cre=1e-3,
          optimistic=False, use_cls_err=False):
    """kind: 'sr' | 'actor' | 'critic'.  r: list per switch of [n, n_sinks]; c_err / d_cor: [n_leaves, n];
    ops: per node n_ops + router.n_ops; k_cpt: scalar or [n].
    Returns dict(p_tr, p_ev [n_nodes, n], loss terms, dr (list), w_cerr [n_leaves, n], node_stat)."""
    T = lambda a: torch.as_tensor(np.asarray(a, np.float64))
    n = np.asarray(c_err).shape[1]
    rs = [T(x).clone().requires_grad_(True) for x in r]
    ce = T(c_err).clone().requires_grad_(True)
    dc = T(d_cor)
    kc = T(k_cpt) * torch.ones(n, dtype=torch.float64)
    ones = torch.ones(n, dtype=torch.float64)
    N = tree.nodes
    root_leaves = tree.n_leaves(0)
    p_ϵ = lambda i: ϵ * tree.n_leaves(i) / root_leaves
    P_tr, P_ev, C_ev, C_opt, C_cre = {}, {}, {}, {}, {}

    def own_err(i):
        lf = N[i].get('leaf_id') if not N[i]['sinks'] else None
        if use_cls_err and kind == 'critic':
            return (1 - dc[lf]) if lf is not None else torch.zeros(n, dtype=torch.float64)
        return ce[lf] if lf is not None else torch.zeros(n, dtype=torch.float64)

    def walk(i, p_tr, p_ev):
        P_tr[i], P_ev[i] = p_tr, p_ev
        sinks = N[i]['sinks']
        if len(sinks) < 2 or kind == 'sr':
            for c in sinks:
                walk(c, p_tr, p_ev)
            C_ev[i] = own_err(i) + kc * ops[i] + sum(C_ev[c] for c in sinks)
            C_opt[i] = own_err(i) + kc * ops[i] + sum(C_opt[c] for c in sinks)
            C_cre[i] = 0.0
            return
        rx = rs[N[i]['switch_id']]
        π_tr = ((1 - p_ϵ(i) / p_tr[:, None]) * torch.softmax(rx / τ, 1)
                + torch.tensor([p_ϵ(c) for c in sinks], dtype=torch.float64) / p_tr[:, None])
        π_ev = (torch.argmax(rx, 1)[:, None] == torch.arange(len(sinks))).to(torch.float64)
        for k, c in enumerate(sinks):
            walk(c, p_tr * π_tr[:, k], p_ev * π_ev[:, k])
        C_ev[i] = own_err(i) + kc * ops[i] + sum(π_ev[:, k] * C_ev[c] for k, c in enumerate(sinks))
        C_opt[i] = own_err(i) + kc * ops[i] + reduce(torch.minimum, (C_opt[c] * ones for c in sinks))
        C_cre[i] = k_cre * sum((rx[:, k] + (C_opt[c] if optimistic else C_ev[c]).detach()) ** 2
                               for k, c in enumerate(sinks))
    walk(0, ones, ones)
    idx = range(len(N))
    leaf_err = lambda i: ce[N[i]['leaf_id']] if not N[i]['sinks'] else 0.0
    if kind == 'sr':
        l_err, l_cpt, l_aux = sum(leaf_err(i) * ones for i in idx), 0 * ones, 0 * ones
    elif kind == 'actor':
        l_err = sum(P_tr[i] * leaf_err(i) for i in idx)
        l_cpt = sum(P_tr[i] * kc * ops[i] for i in idx)
        l_aux = sum(P_tr[i].detach() * k_dec * (rs[N[i]['switch_id']] ** 2).sum(1) for i in tree.switches) + 0 * ones
    else:
        l_err = sum(P_tr[i].detach() * leaf_err(i) for i in idx)
        l_cpt = 0 * ones
        l_aux = sum(P_tr[i].detach() * C_cre[i] for i in idx) + 0 * ones
    c_tot = (l_err + l_cpt + l_aux).mean()
    c_tot.backward()
    z = lambda t: t.grad.numpy() if t.grad is not None else np.zeros(tuple(t.shape))
    P = np.stack([P_tr[i].detach().numpy() for i in idx])
    return dict(p_tr=P, p_ev=np.stack([P_ev[i].detach().numpy() for i in idx]),
                loss=np.array([float(l_err.sum()), float(l_cpt.sum()), float(l_aux.sum()), n]),
                dr=[z(t) for t in rs], w_cerr=z(ce),
                node_stat=np.stack([P.sum(1), (P ** 2).sum(1)], 1))
