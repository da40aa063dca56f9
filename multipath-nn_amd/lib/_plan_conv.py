"""Execution plan of a plain single-scale conv net: SURVEY 8 row a10.

The shipped specs never instantiate ``Conv`` (scripts/lib/layer_types.py:55-74), but it is part of
the operator surface north_star names ("3x3 / 1x1 contractions").  This engine runs the nets one
can build from it with the reference's protocol -- a statically-routed net whose root is a chain of
``Conv`` layers (supp 3 or 1, optional ``res`` identity initialisation), each optionally followed by
``Rect``, optionally interleaved with ``MaxPool`` / ``GlobalMaxPool`` (layer_types.py:86-100: max-pooling commutes with
the ReLU the consumers apply on load, so the engine pools the stored pre-activation maps -- csrc/pool.hip), with a
``LinTrans -> Softmax -> CrossEntropyError`` leaf::

    SRNet(x0_shape=.., y_shape=.., root=Chain(comps=[Conv(n_chan=16, supp=3), Rect(),
                                                     Conv(n_chan=32, supp=1, res=False), Rect()],
                                              sinks=[Chain(comps=[LinTrans(n_chan=10), Softmax(),
                                                                  CrossEntropyError()])]))

on ``mpnn_conv_nhwc_{fwd,dgrad,wgrad}`` (csrc/conv_nhwc.hip), the exit kernels and the optimizer of
the multiscale path.  ReLU is applied by the consumer while loading (MPNN_ACT_RELU) and masked in the
input-gradient epilogue, so -- as on the multiscale path -- only pre-activation maps are stored.
Same surface as lib/_plan.Engine where callers touch it (run / state / P, A, G / init_params).
Launches are eager (this is not the benchmarked path).
"""
import ctypes as C
import os

import numpy as np
import torch

from lib import _hip
from lib.layer_types import Chain
from lib.net_types import params_list_rec
from lib._plan import _attr, OPT_CHUNK


def is_conv_net(net):
    root = net.root
    if net._net_kind != 'sr' or not isinstance(root, Chain) or len(root.sinks) != 1 or not root.comps:
        return False
    names = [type(c).__name__ for c in root.comps]
    if names[0] != 'Conv' or any(n not in ('Conv', 'Rect', 'MaxPool', 'GlobalMaxPool') for n in names):
        return False
    if any(a == 'Rect' and b == 'Rect' for a, b in zip(names, names[1:])):
        return False
    head = root.sinks[0]
    return (isinstance(head, Chain) and not head.sinks and
            [type(c).__name__ for c in head.comps] == ['LinTrans', 'Softmax', 'CrossEntropyError'])


class _Node:
    pass


class ConvEngine:
    def __init__(self, net, device=None, n_max=128):
        self.net = net
        self.lib = _hip.load()
        if not torch.cuda.is_available():
            raise _hip.HipError('no GPU visible: the multipath-nn hot path runs on MI355X only')
        self.dev = torch.device(device or 'cuda:%d' % int(os.environ.get('LOCAL_RANK', '0')))
        torch.cuda.set_device(self.dev)
        self.world, self.allreduce = 1, None
        root, head = net.root, net.root.sinks[0]
        self.x0_shape = tuple(net.hypers.x0_shape)
        self.n_cls = int(net.hypers.y_shape[0])
        # stages: every Conv / MaxPool / GlobalMaxPool of the chain produces a stored map; a Rect only marks the map in
        # front of it as "ReLU on load" (pooling keeps the mark: max-pool and ReLU commute).
        # stage = (layer, relu_on_load_of_its_output, kind, (H_in, W_in), (H_out, W_out))
        self.stages = []
        h, w = self.x0_shape[:2]
        self.C = [self.x0_shape[2]]
        relu = False
        for c in root.comps:
            kind = type(c).__name__
            if kind == 'Rect':
                relu = True
                if self.stages:
                    st = self.stages[-1]
                    self.stages[-1] = (st[0], True) + st[2:]
                continue
            if kind == 'Conv':
                relu = False
                if c.hypers.supp not in (1, 3):
                    raise NotImplementedError('Conv supp %r: the kernels cover 3x3 and 1x1' % (c.hypers.supp,))
                if c.hypers.supp == 3 and (c.params.w.shape[3] % 16 or (c.params.w.shape[2] > 4 and c.params.w.shape[2] % 4)):
                    raise NotImplementedError('3x3 Conv needs a multiple of 16 output channels (and of 4 input channels beyond an image)')
                if c.hypers.supp == 3 and (h != w or h not in (4, 8) and (w % 16 or h % 4)):
                    raise NotImplementedError('3x3 Conv on a %dx%d map: the body covers 4x4, 8x8 and W %% 16 == 0' % (h, w))
                self.stages.append((c, False, 'conv', (h, w), (h, w)))
                self.C.append(c.hypers.n_chan)
            else:
                if kind == 'MaxPool':
                    # the reference calls tf.nn.max_pool(x, strides, k_shape, 'SAME') -- hypers in the order (strides,
                    # k_shape) where TensorFlow expects (ksize, strides): the WINDOW is hypers.stride, the STEP hypers.supp
                    win, step = int(c.hypers.stride), int(c.hypers.supp)
                    ho, wo = -(-h // step), -(-w // step)
                else:
                    win = step = 0
                    ho = wo = 1
                self.stages.append((c, relu, 'gpool' if kind == 'GlobalMaxPool' else 'pool', (h, w), (ho, wo), win, step))
                self.C.append(self.C[-1])
                h, w = ho, wo
        self.H, self.W = h, w                      # the map the head reads
        K = h * w * self.C[-1]
        # beyond the tuned exit kernels' limits (<= 16 classes, C <= 128, K % 16 == 0): the any-width forms (csrc/exit_gen.hip)
        self.generic = self.C[-1] > 128 or K % 16 != 0 or self.n_cls > 16
        if self.generic and (self.lib.mpnn_exit_gen_check(self.C[-1], K, self.n_cls, 0, 0, 0) or self.C[-1] % 4 or 256 % (self.C[-1] // 4)):
            raise NotImplementedError('head on a %dx%dx%d map / %d classes: outside the exit kernels\' limits' % (h, w, self.C[-1], self.n_cls))
        self.head = head
        # routing tree: root chain (node 0) -> leaf (node 1)
        self.nodes = []
        for i, ℓ in enumerate((root, head)):
            nd = _Node()
            nd.idx, nd.layer, nd.parent, nd.sink_index = i, ℓ, i - 1, 0
            self.nodes.append(nd)
        self.nodes[1].leaf_id = 0
        self.leaves, self.switches, self.blocks = [self.nodes[1]], [], []
        self.max_sinks = 2
        self.node_ops_host = [float(root.n_ops), float(head.n_ops)]
        self._alloc()
        self.init_params(net.hypers.__dict__.get('seed'))
        self.n_max = 0
        self._ensure_capacity(n_max)
        self.last_n, self.last_mode = 0, 'ev'

    # ------------------------------------------------------------------ parameters
    def _alloc(self):
        dev, net = self.dev, self.net
        owner = {}
        for nd in self.nodes:
            for p in params_list_rec(nd.layer):
                owner[id(p)] = nd.idx
        self.trainable = [p for p in net._all_params if p.trainable]
        self.state_params = []
        off = 0
        for p in self.trainable:
            off = (off + 3) // 4 * 4               # 16-byte aligned tensors (float4 paths of the slab reduction)
            p.offset, p.node, p.is_router = off, owner[id(p)], 0
            off += p.size
        off = (off + 3) // 4 * 4
        self.n_params = off
        self.P, self.A = torch.zeros(off, device=dev), torch.zeros(off, device=dev)
        # one zero arena: loss | reduction scratch | G (+ node statistics)
        self.cmax = max(self.C)
        zb = 32 + 16 * self.cmax * 16
        self._zarena = torch.zeros(zb + (4 * (off + 4) + 15) // 16 * 16, dtype=torch.uint8, device=dev)
        self.loss = self._zarena[:32].view(torch.float64)
        self.scratch = self._zarena[32:zb].view(torch.float64)
        self.G = self._zarena[zb:].view(torch.float32)[:off + 4]
        self.node_stat = self.G[off:]
        self.S = torch.zeros(1, device=dev)
        for p in self.trainable:
            p.data, p.grad, p.accum = (t[p.offset:p.offset + p.size] for t in (self.P, self.G, self.A))
        seg, eqs, eq_off = [], [], 0
        for p in self.trainable:
            l2 = int(np.float32(p.l2).view(np.int32))
            has_eq = bool(p.l2) and p.eq is not None
            for s in range(0, p.size, OPT_CHUNK):
                seg += [p.offset + s, min(OPT_CHUNK, p.size - s), p.node, 0, l2, eq_off + s if has_eq else -1, 0, 0, 0, -1, -1, 0]
            if has_eq:
                eqs.append(np.asarray(p.eq, np.float32).reshape(-1))
                eq_off += p.size
        self.n_seg = len(seg) // _hip.SEG_INTS
        self.seg = torch.tensor(seg, dtype=torch.int32, device=dev)
        self.w_eq = torch.from_numpy(np.concatenate(eqs)).to(dev) if eqs else None
        # weight packs of the 3x3 stages
        desc, poff = [], 0
        self.pack = {}
        for k, st_ in enumerate(self.stages):
            c = st_[0]
            if st_[2] == 'conv' and c.hypers.supp == 3:
                ci, co = c.params.w.shape[2], c.params.w.shape[3]
                fs = 9 * ((ci + 15) // 16) * 16 * co
                bs = 9 * ((co + 15) // 16) * 16 * ci if ci % 16 == 0 else 0
                desc += [c.params.w.offset, poff, poff + fs if bs else -1, ci, co, 0]
                self.pack[k] = (poff, poff + fs if bs else None)
                poff += fs + bs
        self.n_pack = len(desc) // 6
        self.packs = torch.zeros(max(poff, 1), device=dev)
        self.pack_desc = torch.tensor(desc if desc else [0] * 6, dtype=torch.int32, device=dev)
        self.node_tab = torch.tensor([-1, 0, 1, -1, -1, 1, 0, 0, 0, 0, 0, -1, 0, 1, 1, 1], dtype=torch.int32, device=dev)
        self.kid_tab = torch.zeros(2, dtype=torch.int32, device=dev)
        self.node_ops = torch.tensor(self.node_ops_host, dtype=torch.float32, device=dev)
        self.hyp = torch.zeros(_hip.HYP_N, device=dev)

    def init_params(self, seed=None):
        rng = np.random.default_rng(seed)
        P = np.zeros(self.n_params, np.float32)
        for p in self.net._all_params:
            kind, scale = p.init
            v = (scale * rng.standard_normal(p.size)).astype(np.float32) if kind == 'normal' else \
                (np.ones(p.size, np.float32) if kind == 'ones' else np.zeros(p.size, np.float32))
            if kind == 'normal' and p.eq is not None:
                v = v + np.asarray(p.eq, np.float32).reshape(-1)
            P[p.offset:p.offset + p.size] = v
        self.P.copy_(torch.from_numpy(P))
        self.A.zero_()

    def _ensure_capacity(self, n, train=True):
        if n <= self.n_max:
            return
        self.n_max = n
        z = lambda *s: torch.zeros(s, device=self.dev)
        self.x0, self.y = z(n, *self.x0_shape), z(n, self.n_cls)
        self.out = [z(n, st[4][0], st[4][1], c) for st, c in zip(self.stages, self.C[1:])]     # pre-activation map of every stage
        self.g = [z(n, st[4][0], st[4][1], c) for st, c in zip(self.stages, self.C[1:])]       # gradient w.r.t. it
        self.gcnt = {k: z(n, self.C[k + 1]) for k, st in enumerate(self.stages) if st[2] == 'gpool'}   # maxima per (sample, channel)
        self.z, self.dz = z(n, self.n_cls), z(n, self.n_cls)
        self.c_err, self.d_cor, self.w_cerr = z(n), z(n), z(n)
        self.p_tr, self.p_ev = z(2 * n), z(2 * n)
        self.r, self.dr = z(2 * n), z(2 * n)
        self.n_split = {}
        for k, st in enumerate(self.stages):
            if st[2] == 'conv' and st[0].hypers.supp == 3:
                self.n_split[k] = max(1, min(64, self.lib.mpnn_wgrad_tiles(n, st[3][0], st[3][1])))
        self.slab = z(max(list(self.n_split.values()) + [1]) * max(p.size for p in self.trainable) * 2 + 1024)

    # ------------------------------------------------------------------ running
    def _act(self, k):
        """The input of stage k as its consumers load it (k = len(stages): the head's input)."""
        if k == 0:
            return _hip.act(self.x0, self.C[0], _hip.ACT_IDENTITY)
        return _hip.act(self.out[k - 1], self.C[k], _hip.ACT_RELU if self.stages[k - 1][1] else _hip.ACT_IDENTITY)

    def _chk(self, code, what):
        _hip.check(code, what)

    def run(self, feed, train, routed=False):
        net, lib = self.net, self.lib
        st = torch.cuda.current_stream().cuda_stream
        x0 = feed[net.x0]
        n = int(x0.shape[0])
        mode = feed.get(net.mode, net.mode.default)
        if train != (mode == 'tr'):
            raise ValueError("net.train.run feeds mode 'tr'; forward-only runs evaluate in mode 'ev'")
        self._ensure_capacity(n)
        put = lambda dst, src: dst.copy_(src.reshape(dst.shape) if isinstance(src, torch.Tensor) else
                                         torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32)).reshape(dst.shape))
        put(self.x0[:n], x0); put(self.y[:n], feed[net.y])
        ϕ = net.hypers
        h = torch.zeros(_hip.HYP_N)
        h[_hip.HYP_LR] = float(feed.get(_attr(net, 'λ_lrn'), _attr(ϕ, 'λ_lrn', 0.0)))
        h[_hip.HYP_MU] = float(feed.get(_attr(net, 'μ_lrn'), _attr(ϕ, 'μ_lrn', 0.0)))
        h[_hip.HYP_TAU] = 1.0
        self.hyp.copy_(h)
        keep = []
        z = self._zarena
        if self.n_pack:
            self._chk(lib.mpnn_step_begin(self.P.data_ptr(), self.packs.data_ptr(), self.pack_desc.data_ptr(), self.n_pack,
                                          z.data_ptr(), z.numel(), st), 'step_begin')
        else:
            z.zero_()
        HW = self.H * self.W
        # ---- forward ----
        for k, st_ in enumerate(self.stages):
            c, kind, (hi, wi) = st_[0], st_[2], st_[3]
            if kind == 'conv':
                a = _hip.ConvNhwcFwdArgs()
                a.a = self._act(k)
                a.w = self.packs[self.pack[k][0]:].data_ptr() if c.hypers.supp == 3 else c.params.w.data.data_ptr()
                a.bias, a.out = c.params.b.data.data_ptr(), self.out[k].data_ptr()
                a.n, a.H, a.W, a.Cout, a.supp = n, hi, wi, self.C[k + 1], c.hypers.supp
                self._chk(lib.mpnn_conv_nhwc_fwd(C.byref(a), st), 'conv_nhwc_fwd')
            else:                                  # MaxPool / GlobalMaxPool of the stored pre-activation map
                src = self.x0 if k == 0 else self.out[k - 1]
                cnt = self.gcnt[k].data_ptr() if kind == 'gpool' else None
                self._chk(lib.mpnn_maxpool_fwd(src.data_ptr(), self.out[k].data_ptr(), cnt, n, hi, wi, self.C[k],
                                               st_[5] if kind == 'pool' else 0, st_[6] if kind == 'pool' else 0,
                                               1 if kind == 'gpool' else 0, st), 'maxpool_fwd')
        lt, ce = self.head.comps[0], self.head.comps[2]
        a_head = self._act(len(self.stages))
        if train:
            lf, tf = _hip.LinFwdArgs(), _hip.ExitTailArgs()
            lf.a, lf.HW, lf.n = a_head, HW, n
            lf.w[0], lf.b[0], lf.y[0], lf.M[0] = lt.params.w.data.data_ptr(), lt.params.b.data.data_ptr(), self.z.data_ptr(), self.n_cls
            tf.z, tf.y, tf.n_cls, tf.eps_ce = self.z.data_ptr(), self.y.data_ptr(), self.n_cls, float(ce.hypers.ϵ)
            tf.c_err, tf.d_cor, tf.mode, tf.n = self.c_err.data_ptr(), self.d_cor.data_ptr(), _hip.ACT_BN_BATCH, n
            t_lf, t_tf = _hip.to_device_table([lf], self.dev), _hip.to_device_table([tf], self.dev)
            keep += [t_lf, t_tf]
            self._chk((lib.mpnn_lin_fwd_gen if self.generic else lib.mpnn_lin_fwd)(t_lf.data_ptr(), 1, n, st), 'lin_fwd')
            self._chk((lib.mpnn_exit_tail_fwd_gen if self.generic else lib.mpnn_exit_tail_fwd)(t_tf.data_ptr(), 1, n, st), 'exit_tail_fwd')
        else:
            e = _hip.ExitEvArgs()
            e.a, e.HW, e.n = a_head, HW, n
            e.w_head, e.b_head, e.n_cls = lt.params.w.data.data_ptr(), lt.params.b.data.data_ptr(), self.n_cls
            e.y, e.eps_ce, e.c_err, e.d_cor = self.y.data_ptr(), float(ce.hypers.ϵ), self.c_err.data_ptr(), self.d_cor.data_ptr()
            if not self.generic:
                self._chk(lib.mpnn_exit_ev_check(C.byref(e)), 'exit_ev record')
            else:
                e.z = self.z.data_ptr()            # (scratch map of mpnn_exit_ev_gen: the head logits)
            t_e = _hip.to_device_table([e], self.dev)
            keep.append(t_e)
            self._chk((lib.mpnn_exit_ev_gen if self.generic else lib.mpnn_exit_ev)(t_e.data_ptr(), 1, n, st), 'exit_ev')
        ra = _hip.RouteArgs()
        ra.net_type, ra.n_nodes, ra.n_leaves, ra.n_switches, ra.max_sinks = _hip.NET_SR, 2, 1, 0, 2
        ra.want_grad = 1 if train else 0
        ra.nodes, ra.sw_children, ra.node_ops, ra.hyp = self.node_tab.data_ptr(), self.kid_tab.data_ptr(), self.node_ops.data_ptr(), self.hyp.data_ptr()
        ra.r, ra.c_err, ra.d_cor = self.r.data_ptr(), self.c_err.data_ptr(), self.d_cor.data_ptr()
        ra.p_tr, ra.p_ev, ra.w_cerr, ra.dr = self.p_tr.data_ptr(), self.p_ev.data_ptr(), self.w_cerr.data_ptr(), self.dr.data_ptr()
        ra.node_stat, ra.loss, ra.n, ra.n_total = (self.node_stat.data_ptr() if train else None), self.loss.data_ptr(), n, n
        self._chk(lib.mpnn_route(C.byref(ra), st), 'route')
        if train:
            self._backward(n, a_head, lf, tf, keep, st)
        torch.cuda.synchronize()               # (argument records above are host objects of this call)
        self.last_n, self.last_mode = n, mode
        root, head = self.net.root, self.head
        root.p_tr, root.p_ev = self.p_tr[:n], self.p_ev[:n]
        head.p_tr, head.p_ev = self.p_tr[n:2 * n], self.p_ev[n:2 * n]
        head.c_err, head.δ_cor = self.c_err[:n], self.d_cor[:n]

    def _backward(self, n, a_head, lf, tf, keep, st):
        lib = self.lib
        HW, last = self.H * self.W, len(self.stages) - 1
        lt = self.head.comps[0]
        tb, lb = _hip.ExitTailBwdArgs(), _hip.LinBwdArgs()
        tb.f, tb.w_cerr, tb.dz = tf, self.w_cerr.data_ptr(), self.dz.data_ptr()
        lb.a, lb.HW, lb.n = a_head, HW, n
        lb.w[0], lb.dy[0], lb.M[0] = lf.w[0], self.dz.data_ptr(), self.n_cls
        lb.dw[0], lb.db[0] = lt.params.w.grad.data_ptr(), lt.params.b.grad.data_ptr()
        masked = self.stages[last][1]          # Rect before the head: masked dX + (discarded) reductions
        if masked and not self.generic:
            lb.dz_out, lb.red_out, lb.red_nslot = self.g[last].data_ptr(), self.scratch.data_ptr(), 1
        else:
            lb.dx = self.g[last].data_ptr()
        t_tb, t_lb = _hip.to_device_table([tb], self.dev), _hip.to_device_table([lb], self.dev)
        keep += [t_tb, t_lb]
        self._chk((lib.mpnn_exit_tail_bwd_gen if self.generic else lib.mpnn_exit_tail_bwd)(t_tb.data_ptr(), 1, n, st), 'exit_tail_bwd')
        self._chk((lib.mpnn_lin_bwd_gen if self.generic else lib.mpnn_lin_bwd)(t_lb.data_ptr(), 1, n, HW * self.C[-1], st), 'lin_bwd')
        if masked and self.generic:
            # the any-width lin_bwd leaves the plain dX: the ReLU mask is mpnn_bn_bwd_reduce with a plain-ReLU context
            # (dz = dy where the stored pre-activation is positive; its reductions go to scratch), in place
            ctx = _hip.BnCtx()
            ctx.s = self.out[last].data_ptr()
            ctx.bn = _hip.act(None, self.C[-1], _hip.ACT_RELU, 0, None, n * HW)
            ctx.red_nslot = 1
            keep.append(ctx)
            self._chk(lib.mpnn_bn_bwd_reduce(self.g[last].data_ptr(), C.byref(ctx), self.g[last].data_ptr(), self.scratch.data_ptr(),
                                             n * HW, st), 'relu mask of the head\'s dX')
        for k in range(last, -1, -1):
            st_ = self.stages[k]
            c, kind, (hi, wi) = st_[0], st_[2], st_[3]
            if kind != 'conv':
                if k == 0:
                    break
                cnt = self.gcnt[k].data_ptr() if kind == 'gpool' else None
                self._chk(lib.mpnn_maxpool_bwd(self.out[k - 1].data_ptr(), self.out[k].data_ptr(), cnt, self.g[k].data_ptr(),
                                               self.g[k - 1].data_ptr(), n, hi, wi, self.C[k],
                                               st_[5] if kind == 'pool' else 0, st_[6] if kind == 'pool' else 0,
                                               1 if kind == 'gpool' else 0, st), 'maxpool_bwd')
                continue
            supp = c.hypers.supp
            w = _hip.ConvNhwcWgradArgs()
            w.a, w.g = self._act(k), self.g[k].data_ptr()
            w.n, w.H, w.W, w.Cout, w.supp = n, hi, wi, self.C[k + 1], supp
            pw, pb = c.params.w, c.params.b
            # (n_split was sized for the CAPACITY; a smaller batch may have fewer pixel tiles than that, and a split without
            # a tile never writes its slab -- the reduction would add whatever an earlier stage left there)
            split = min(self.n_split[k], max(1, lib.mpnn_wgrad_tiles(n, hi, wi))) if supp == 3 else 1
            if supp == 3 and split > 1:
                stride = (pw.size + pb.size + 3) // 4 * 4
                w.dw, w.db = self.slab.data_ptr(), self.slab[pw.size:].data_ptr()
                w.n_split, w.split_stride = split, stride
            else:
                w.dw, w.db, w.n_split = pw.grad.data_ptr(), pb.grad.data_ptr(), 1
            self._chk(lib.mpnn_conv_nhwc_wgrad(C.byref(w), st), 'conv_nhwc_wgrad')
            if supp == 3 and split > 1:
                tab = []
                for prm, o in ((pw, 0), (pb, pw.size)):
                    item = _hip.slab_item_size(split)
                    for j in range(0, prm.size, item):
                        tab += [o + j, prm.offset + j, min(item, prm.size - j), split, stride, 0]
                t = torch.tensor(tab, dtype=torch.int32, device=self.dev)
                keep.append(t)
                self._chk(lib.mpnn_slab_reduce(self.slab.data_ptr(), self.G.data_ptr(), t.data_ptr(), len(tab) // 6, st), 'slab_reduce')
                torch.cuda.current_stream().synchronize()      # the slab is reused by the next stage
            if k == 0:
                break
            d = _hip.ConvNhwcDgradArgs()
            d.g, d.Cg = self.g[k].data_ptr(), self.C[k + 1]
            d.w = self.packs[self.pack[k][1]:].data_ptr() if supp == 3 else pw.data.data_ptr()
            if self.stages[k - 1][1]:
                d.relu_src, d.scratch = self.out[k - 1].data_ptr(), self.scratch.data_ptr()
            d.dx = self.g[k - 1].data_ptr()
            d.n, d.H, d.W, d.Cin, d.supp = n, hi, wi, self.C[k], supp
            if supp == 3 and self.pack[k][1] is None:
                raise NotImplementedError('3x3 Conv above the first stage needs a multiple of 16 input channels')
            self._chk(lib.mpnn_conv_nhwc_dgrad(C.byref(d), st), 'conv_nhwc_dgrad')
        if self.allreduce is not None:
            # data parallel (lib/_dp.py): every rank's gradient SUMS over its n samples -> all-reduce (sum), then the
            # optimizer divides by n * world: the step of the global batch (no BatchNorm here: nothing else to exchange)
            h = self.allreduce(self.G)
            if hasattr(h, 'wait'):
                h.wait()
        self._chk(lib.mpnn_talr_momentum_step(self.P.data_ptr(), self.A.data_ptr(), self.G.data_ptr(), self.seg.data_ptr(),
                                              self.n_seg, self.node_stat.data_ptr(), self.hyp.data_ptr(), 0, 1.0 / (n * self.world), 1.0 / self.world,
                                              self.w_eq.data_ptr() if self.w_eq is not None else None, None, st), 'talr_momentum_step')

    def state(self):
        net, n = self.net, self.last_n
        ℓ, y = self.head, self.y[:n]
        return {(net, 'acc'): ℓ.p_ev * ℓ.δ_cor,
                (net, 'moc'): net.root.p_ev * self.node_ops_host[0] + ℓ.p_ev * self.node_ops_host[1],
                (ℓ, 'p_cor'): ℓ.p_ev * ℓ.δ_cor, (ℓ, 'p_inc'): ℓ.p_ev * (1 - ℓ.δ_cor),
                (ℓ, 'p_cor_by_cls'): (ℓ.p_ev * ℓ.δ_cor)[:, None] * y,
                (ℓ, 'p_inc_by_cls'): (ℓ.p_ev * (1 - ℓ.δ_cor))[:, None] * y, (ℓ, 'c_err'): ℓ.c_err}
