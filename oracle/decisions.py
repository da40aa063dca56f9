"""The product's discrete decisions of one training step, for the decision-forced oracle run.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  A training step contains two kinds of discrete
choice -- which element of a 2x2 window is the maximum (layer_types.py:109-110,185) and on which
side of zero a BatchNorm output falls (ReLU, layer_types.py:76-79,196-199).  fp32 kernels and the
float64 oracle disagree on a near-tie now and then, and one flipped element changes some gradient
tensors by tens of percent.  ``from_product`` reads the decisions the DEVICE took (pool arg-max
from the pre-BatchNorm maps it stored; ReLU sides from ``mpnn_bn_relu_fwd``, the materialised form
of the activation every consumer applies on load) so that ``RefNet.forward(forced=...)`` can
differentiate the same piecewise-linear branch: every tensor must then agree to rounding.
"""
import ctypes as C

import numpy as np
import torch

from oracle import np_ops as O


def from_product(net, n, before):
    """before: {id(Param): tensor} of the parameters the step started from (the BatchNorm gamma /
    beta the forward pass used).  Call after ``net.train.run`` (the forward statistics are still in
    the engine's arena)."""
    from lib import _hip
    eng = net.engine()
    lib = eng.lib
    st = torch.cuda.current_stream().cuda_stream
    forced = {}
    for b in eng.blocks:
        chain, conv = b.node.layer, b.conv
        masks = []
        for i in range(b.L):
            s = b.s[i][:n]
            if i < b.L - 1:
                forced[('pool', id(conv), i)] = O.pool2_argfirst(s.cpu().numpy())
            bn = b.bns[i].params
            act = _hip.act(s, b.C[i], _hip.ACT_BN_BATCH, 0,
                           dict(sum=eng.batch_stat_sums()[b.sum_off[i]:], gamma=before[id(bn.γ)], beta=before[id(bn.β)],
                                m_avg=bn.m_avg.data, v_avg=bn.v_avg.data, eps=float(b.bns[i].hypers.ϵ),
                                nslot=eng._nslot(b, i)), n * b.H[i] * b.W[i])
            y = torch.empty_like(s)
            _hip.check(lib.mpnn_bn_relu_fwd(C.byref(act), y.data_ptr(), n * b.H[i] * b.W[i], st), 'bn_relu_fwd')
            masks.append((y > 0).cpu().numpy())
        forced[('relu', id(chain), 2)] = masks
        if b.router is not None:
            R, R2 = b.R, b.R2              # bn_save = m1 [R], rstd1 [R], m2 [R2], rstd2 [R2]
            sv = b.bn_save.cpu().numpy()
            rc = b.router.comps
            f32 = np.float32
            for k, h, (m, rstd) in ((3, b.h1, (sv[:R], sv[R:2 * R])), (6, b.h2, (sv[2 * R:2 * R + R2], sv[2 * R + R2:2 * R + 2 * R2]))):
                bn = rc[k - 1].params
                g, be = before[id(bn.γ)].cpu().numpy().astype(f32), before[id(bn.β)].cpu().numpy().astype(f32)
                hv = h[:n].cpu().numpy().astype(f32)
                a = (g * (hv - m.astype(f32))) * rstd.astype(f32) + be          # exit_tail.hip, same order
                forced[('relu', id(b.router), k)] = a > 0
    return forced
