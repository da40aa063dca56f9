"""Tree classification and the flat HBM buffers of a net: parameters / momentum / gradients in the order the backward
pass finishes them, BatchNorm state, the fp64 statistic arenas, per-block activation and gradient maps sized for a
batch capacity."""
import ctypes as C
import os

import numpy as np
import torch

from lib import _hip
from lib.layer_types import Chain
from lib.net_types import n_leaves, params_list_rec
from lib._eng_common import (BLOCK_COMPS, CAPTURE_MODE, HEAD_COMPS, OPT_CHUNK, ROUTER_COMPS, BoundInput, _attr, _Block, _kind, _nf,
                             _Node)


class Allocation:

    # ------------------------------------------------------------------ structure
    def _classify(self):
        net = self.net
        self.nodes = []
        index = {}
        for ℓ in net.layers:
            nd = _Node()
            nd.idx, nd.layer, nd.kind = len(self.nodes), ℓ, _kind(ℓ)
            nd.parent, nd.sink_index = -1, 0
            index[id(ℓ)] = nd
            self.nodes.append(nd)
        for nd in self.nodes:
            for i, s in enumerate(nd.layer.sinks):
                index[id(s)].parent, index[id(s)].sink_index = nd.idx, i
        self.leaves = [nd for nd in self.nodes if len(nd.layer.sinks) == 0]
        self.switches = [nd for nd in self.nodes if len(nd.layer.sinks) > 1]
        for i, nd in enumerate(self.leaves):
            nd.leaf_id = i
        for i, nd in enumerate(self.switches):
            nd.switch_id = i
        self.max_sinks = max([len(nd.layer.sinks) for nd in self.switches] + [2])
        if len(self.nodes) > _hip.MAX_NODES or self.max_sinks > _hip.MAX_SINKS:
            raise NotImplementedError('routing tree too large for mpnn_route')
        kind = self.net._net_kind
        for nd in self.nodes:
            r = nd.layer.router
            if r is not None:
                if kind == 'sr' or len(nd.layer.sinks) < 2:
                    raise NotImplementedError('router on a node with < 2 sinks / in an SRNet')
                if not isinstance(r, Chain) or [type(c).__name__ for c in r.comps] != ROUTER_COMPS:
                    raise NotImplementedError('router chain outside the MI355X hot path')
                if nd.kind != 'block':
                    raise NotImplementedError('router on a %s node' % nd.kind)
            elif len(nd.layer.sinks) > 1 and kind != 'sr':
                raise NotImplementedError('switch without router')
            if nd.kind == 'head' and nd.layer.sinks:
                raise NotImplementedError('LogReg with sinks')
        root = self.nodes[0]
        if root.kind != 'pyramid':
            raise NotImplementedError('root must be the ToPyramid chain')
        self.x0_shape = tuple(self.net.hypers.x0_shape)
        self.n_cls = int(self.net.hypers.y_shape[0])
        # blocks
        self.blocks = []
        self.generic_exits = bool(int(os.environ.get('MPNN_GENERIC_EXITS', '0')))      # (1: the any-width exit kernels for every net)
        for nd in self.nodes:
            if nd.kind != 'block':
                continue
            b = _Block()
            b.node = nd
            conv, mbn, _ = nd.layer.comps
            b.conv, b.bns = conv, mbn.comps
            b.L = len(conv.hypers.n_chan)
            b.H = [s.shape[0] for s in conv.x]
            b.W = [s.shape[1] for s in conv.x]
            b.C = [s.shape[2] for s in conv.x]
            for h, w in zip(b.H, b.W):
                if h != w:
                    raise NotImplementedError('non-square feature maps')
            for i in range(b.L):
                if tuple(getattr(conv.params, 'w_horz_%i' % i).shape[:2]) != (3, 3):
                    raise NotImplementedError('only 3x3 filters are on the hot path')
            par = self.nodes[nd.parent]
            b.parent = getattr(par, 'block', None)
            if par.kind == 'pyramid':
                n_pyr = par.layer.comps[0].hypers.n_scales
                b.in_shift = [n_pyr - b.L + i for i in range(b.L)]
                b.in_map = None
                b.Cin = [self.x0_shape[2]] * b.L
            elif par.kind == 'block':
                b.in_map = [b.parent.L - b.L + i for i in range(b.L)]
                b.in_shift = [0] * b.L
                b.Cin = [b.parent.C[j] for j in b.in_map]
            else:
                raise NotImplementedError('block below a %s node' % par.kind)
            b.children = []
            nd.block = b
            self.blocks.append(b)
        for b in self.blocks:
            kids = [self.nodes_by_layer(s) for s in b.node.layer.sinks]
            b.children = [k.block for k in kids if k.kind == 'block']       # tree nets: several (arch_and_hypers.py:99-127)
            b.sink_blocks = [k.block if k.kind == 'block' else None for k in kids]
            hs = [k for k in kids if k.kind == 'head']
            if len(hs) > 1:
                raise NotImplementedError('more than one LogReg under a block')
            b.head = hs[0] if hs else None
            b.router = b.node.layer.router
            b.has_exit = b.head is not None or b.router is not None
            # which scales' BN outputs are consumed (by child blocks or the exit)
            b.has_dz = [False] * b.L
            for cb in b.children:
                for j in cb.in_map:
                    b.has_dz[j] = True
            if b.has_exit:
                b.has_dz[b.L - 1] = True
            # compile-time limits of the TUNED exit kernels (exit_tail.hip, exit_ev.hip, lin.hip): <= 16 classes, two
            # equal router layers of <= 16 units, C <= 128 with H*W*C % 16 == 0.  A net with an exit beyond them runs ALL
            # its exits on the any-width forms (csrc/exit_gen.hip: plain kernels, same records), whose own limits are
            # checked here; beyond those the engine refuses instead of truncating.
            if b.has_exit:
                K = b.H[-1] * b.W[-1] * b.C[-1]
                R = R2 = 0
                if b.router is not None:
                    R, R2 = (b.router.comps[k].hypers.n_chan for k in (1, 4))
                    if len(b.node.layer.sinks) > _hip.MAX_SINKS:
                        raise NotImplementedError('more than %d sinks under one switch' % _hip.MAX_SINKS)
                tuned = b.C[-1] <= 128 and K % 16 == 0 and (b.head is None or self.n_cls <= 16) and R == R2 and R <= 16
                if not tuned:
                    self.generic_exits = True
                    if self.lib.mpnn_exit_gen_check(b.C[-1], K, self.n_cls if b.head is not None else 0, R, R2,
                                                    len(b.node.layer.sinks) if b.router is not None else 0):
                        raise NotImplementedError('exit on a %dx%dx%d map with %d classes and a %d-%d router: outside the any-width '
                                                  'exit kernels too (C <= 256, H*W*C <= 4096, <= 1024 classes, <= 256 units)'
                                                  % (b.H[-1], b.W[-1], b.C[-1], self.n_cls, R, R2))
        for nd in self.nodes:
            if nd.kind == 'head' and self.nodes[nd.parent].kind != 'block':
                raise NotImplementedError('LogReg must hang off a ReConvMax block')


    def nodes_by_layer(self, ℓ):
        for nd in self.nodes:
            if nd.layer is ℓ:
                return nd
        raise KeyError(ℓ)


    # ------------------------------------------------------------------ parameters
    def _alloc_params(self):
        owner = {}
        for nd in self.nodes:
            for p in params_list_rec(nd.layer):
                owner[id(p)] = (nd.idx, 0)
            for p in params_list_rec(nd.layer.router):
                owner[id(p)] = (nd.idx, 1)
        # Flat layout in the order the BACKWARD pass finishes the gradients, so that data-parallel
        # training can all-reduce contiguous buckets while later gradients are still being computed:
        #   class 0: exit parameters (heads + routers): final after mpnn_lin_bwd, before the trunk backward
        #   class 1: conv weights / biases, deepest block first (the order the trunk backward runs)
        #   class 2: BatchNorm gamma / beta of the blocks (written by the launch that ends the backward)
        rev = {id(b): k for k, b in enumerate(reversed(self.blocks))}

        def ready_class(p):
            nd = self.nodes[owner[id(p)][0]]
            if owner[id(p)][1] or nd.kind != 'block':
                return (0, 0)
            if type(p.owner).__name__ == 'MultiscaleConvMax':
                return (1, rev[id(nd.block)])
            return (2, 0)
        self.trainable = sorted((p for p in self.net._all_params if p.trainable), key=ready_class)
        self.state_params = [p for p in self.net._all_params if not p.trainable]
        # The per-node TALR statistics (sum p_tr, sum p_tr^2; net_types.py:25-27) sit at the HEAD of G: they are final
        # right after mpnn_route -- before any gradient -- and every segment's learning-rate scale needs them, so under
        # data parallelism they ride in the FIRST bucket and each bucket can be applied as soon as it is reduced.
        # P and A keep the same (unused) prefix: one offset addresses a parameter in all three buffers.
        n_stat = 2 * len(self.nodes)
        off = self.stat_pad = (n_stat + 3) // 4 * 4
        cls_end, blk_end = {0: off, 1: off, 2: off}, {}
        for p in self.trainable:
            # every tensor starts on a 16-byte boundary: the kernels that stream gradients (slab reduction:
            # float4 loads and stores) take a 4x slower scalar path for a misaligned destination, and one
            # 10-float head bias would misalign everything behind it
            off = (off + 3) // 4 * 4
            p.offset, p.node, p.is_router = off, *owner[id(p)]
            off += p.size
            c = ready_class(p)
            for k in range(c[0], 3):
                cls_end[k] = off
            if c[0] == 1:
                blk_end[c[1]] = off
        off = (off + 3) // 4 * 4
        self.n_params = off
        # gradient buckets [lo, hi) in floats of G (the TALR node statistics ride at the head of the first one):
        # exits | conv of the blocks the backward finishes first (>= 40 % of the conv floats) | the rest
        conv_lo, conv_hi = cls_end[0], cls_end[1]
        cut, self.dp_cut_block = conv_hi, None
        for k in sorted(blk_end):
            if blk_end[k] - conv_lo >= 0.4 * (conv_hi - conv_lo) and blk_end[k] < conv_hi:
                cut, self.dp_cut_block = blk_end[k], k          # k: index in reversed(self.blocks)
                break
        end = off
        self.dp_buckets = {}                                   # name -> (lo, hi); markers of the same names in the program
        # ONE bucket by default: the whole of G is all-reduced after the launch that ends the backward pass.  The
        # bucketed form (3: exits | deep blocks | rest, each all-reduce issued where its bucket becomes final, beside the
        # rest of the backward pass) hides two of three collectives, but inside the step's hipGraph every parallel
        # branch that starts in the MIDDLE of the main branch stalls the main branch by ~30 us on this runtime
        # (profiles/r04_dp_corunner.txt: 498 -> 573 us with two 40-us stand-in kernels that overlap perfectly in the
        # kernel trace; no runtime knob changes it, profiles/r04_dp_env_sweep.txt), which is more than a 0.7-0.9 MB
        # all-reduce over xGMI takes.  A branch at the END of the graph (the one-bucket form) costs ~2 us.
        n_buckets = int(os.environ.get('MPNN_DP_BUCKETS', '1'))
        if os.environ.get('MPNN_DP_OVERLAP', '1') == '0':      # no overlap at all: the comparison point of the bucketed form
            n_buckets = 1
        if n_buckets <= 1:                                     # ONE all-reduce of the whole of G after the backward pass
            conv_lo, self.dp_cut_block = 0, None
        elif n_buckets == 2:                                   # exits | everything else
            self.dp_cut_block = None
        if conv_lo > self.stat_pad:
            self.dp_buckets['exit'] = (0, conv_lo)             # (with the node statistics at its head)
        else:
            conv_lo = 0
        if self.dp_cut_block is not None:
            self.dp_buckets['mid'] = (conv_lo, cut)
            self.dp_buckets['end'] = (cut, end)
        else:
            self.dp_buckets['end'] = (conv_lo, end)
        soff = 0
        for p in self.state_params:
            p.offset = soff
            soff += p.size
        dev = self.dev
        self.P = torch.zeros(off, device=dev)
        self.A = torch.zeros(off, device=dev)
        self.G = torch.zeros(off, device=dev)
        self.S = torch.zeros(max(soff, 1), device=dev)
        self.node_stat = self.G[:n_stat]
        for p in self.trainable:
            p.data = self.P[p.offset:p.offset + p.size]
            p.grad = self.G[p.offset:p.offset + p.size]
            p.accum = self.A[p.offset:p.offset + p.size]
        for p in self.state_params:
            p.data = self.S[p.offset:p.offset + p.size]
        # optimizer work items
        # weight packs
        desc, poff, pack_of = [], 0, {}
        for b in self.blocks:
            b.pack = {}
            for i in range(b.L):
                names = ['w_horz_%i' % i] + (['w_vert_%i' % (i - 1)] if i > 0 else [])
                for name in names:
                    p = getattr(b.conv.params, name)
                    ci, co = p.shape[2], p.shape[3]
                    fs = 9 * ((ci + 15) // 16) * 16 * co
                    bs = 9 * ((co + 15) // 16) * 16 * ci if ci % 16 == 0 else 0
                    desc += [p.offset, poff, poff + fs if bs else -1, ci, co, 0]
                    pack_of[id(p)] = (p.offset, ci, co, poff, poff + fs if bs else -1)
                    b.pack[name] = (poff, poff + fs if bs else None)
                    poff += fs + bs
        self.n_pack = len(desc) // 6
        self.packs = torch.zeros(max(poff, 1), device=dev)
        self.pack_desc = torch.tensor(desc, dtype=torch.int32, device=dev)
        # `res` layers (layer_types.py:46,52,65-72): L2 pulls towards w_eq, the identity part of the init
        seg, eqs, eq_off = [], [], 0
        self._seg_owner = []                                # parameter of every optimizer work item
        self._opt_info = {}                                 # id(p) -> (l2 bits, w_eq offset | -1, pack fields)

        for p in self.trainable:
            l2 = np.float32(p.l2).view(np.int32)
            has_eq = bool(p.l2) and p.eq is not None
            pk = pack_of.get(id(p), (0, 0, 0, -1, -1))     # conv weights: the optimizer also refreshes their packs
            self._opt_info[id(p)] = (int(l2), eq_off if has_eq else -1, pk)
            for s in range(0, p.size, OPT_CHUNK):
                seg += [p.offset + s, min(OPT_CHUNK, p.size - s), p.node, p.is_router, int(l2), eq_off + s if has_eq else -1,
                        pk[0], pk[1], pk[2], pk[3], pk[4], 0]
                self._seg_owner.append(id(p))
            if has_eq:
                eqs.append(np.asarray(p.eq, np.float32).reshape(-1))
                eq_off += p.size
        self.w_eq = torch.from_numpy(np.concatenate(eqs)).to(dev) if eqs else None
        self.n_seg = len(seg) // _hip.SEG_INTS
        # optimizer work items of each gradient bucket (the items are in layout order): [first, count)
        seg_off = seg[0::_hip.SEG_INTS]
        self.seg_range = {}
        for name, (lo, hi) in self.dp_buckets.items():
            ks = [k for k, o in enumerate(seg_off) if lo <= o < hi]
            self.seg_range[name] = (ks[0], len(ks)) if ks else (0, 0)
            assert not ks or ks == list(range(ks[0], ks[0] + len(ks)))
        for p in self.trainable:
            p._on_assign = self.invalidate_packs
        self._packs_fresh = False
        self.seg = torch.tensor(seg, dtype=torch.int32, device=dev)
        # fp64 BatchNorm arenas + finalize table
        doff, tab = 0, []
        for b in self.blocks:
            b.sum_off = []
            for i in range(b.L):
                bn = b.bns[i].params
                b.sum_off.append(doff)
                tab += [doff, bn.m_avg.offset, bn.v_avg.offset, b.C[i], b.H[i] * b.W[i],
                        bn.γ.offset if b.has_dz[i] else -1, bn.β.offset, self._nslot(b, i)]
                doff += 2 * b.C[i] * _hip.BN_SLOTS
        # dsum | dred | loss live in ONE byte arena so a step zeroes them with a single memset
        nd = max(doff, 1)
        # ... together with the gradient tensor (and its node-statistics tail): mpnn_step_begin clears
        # the whole arena in the launch that packs the weights.
        zb = (2 * nd + 4) * 8
        gb = (self.G.numel() * 4 + 15) // 16 * 16
        self._zarena = torch.zeros(zb + gb, dtype=torch.uint8, device=dev)
        z64 = self._zarena[:zb].view(torch.float64)
        self.dsum, self.dred, self.loss = z64[:nd], z64[nd:2 * nd], z64[2 * nd:2 * nd + 4]
        # the forward sums of the LAST completed training step (the live ones are cleared by their last reader)
        self.dsum_last = torch.zeros(nd, dtype=torch.float64, device=dev)
        n_g = self.G.numel()
        self.G = self._zarena[zb:zb + n_g * 4].view(torch.float32)
        self.node_stat = self.G[:n_stat]
        for p in self.trainable:
            p.grad = self.G[p.offset:p.offset + p.size]
        self.n_bn = len(tab) // 8
        self.bn_table = torch.tensor(tab, dtype=torch.int32, device=dev)
        # ONE moving-average decay for the conv BatchNorms of a net (a kernel argument of the finishing launch).  Trees
        # built through the layer classes always satisfy this: MultiscaleBatchNorm gives every scale a default
        # BatchNorm() whatever it was handed (reference layer_types.py:246).  A tree whose comps were edited by hand
        # is refused instead of trained with block 0's number.
        decays = sorted({float(bn.hypers.d) for b in self.blocks for bn in b.bns})
        if len(decays) > 1:
            raise NotImplementedError('conv BatchNorms with different moving-average decays %r are outside the MI355X '
                                      'hot path (one decay per net)' % (decays,))
        self.bn_decay = decays[0] if decays else 0.9
        # routing tables
        nodes, ops = [], []
        for nd in self.nodes:                                   # (DFS preorder: a parent comes before its children)
            nd.depth = 0 if nd.parent < 0 else self.nodes[nd.parent].depth + 1
        order = sorted(range(len(self.nodes)), key=lambda j: (self.nodes[j].depth, j))
        rank = {j: k for k, j in enumerate(order)}
        for nd in self.nodes:
            ℓ = nd.layer
            nodes += [nd.parent, nd.sink_index, len(ℓ.sinks), getattr(nd, 'switch_id', -1),
                      getattr(nd, 'leaf_id', -1), n_leaves(ℓ), nd.depth, rank[nd.idx]]
            ops.append(float(ℓ.n_ops + (ℓ.router.n_ops if ℓ.router is not None else 0)))
        kids = []
        for nd in self.switches:
            row = [self.nodes_by_layer(s).idx for s in nd.layer.sinks]
            kids += row + [0] * (self.max_sinks - len(row))
        self.node_tab = torch.tensor(nodes, dtype=torch.int32, device=dev)
        self.kid_tab = torch.tensor(kids if kids else [0], dtype=torch.int32, device=dev)
        self.node_ops = torch.tensor(ops, dtype=torch.float32, device=dev)
        self.node_ops_host = ops
        self.hyp = torch.zeros(_hip.HYP_N, device=dev)
        self._hyp_stage = torch.zeros(_hip.HYP_N)
        self._hyp_ring = [(torch.zeros(_hip.HYP_N).pin_memory(), None) for _ in range(8)]
        self._hyp_slot = -1
        self._hyp_sent = None


    def init_params(self, seed=None):
        """Draw every parameter from the reference's initialisation law
        (layer_types.py:48-50, 64-71, 156-173, 227-230)."""
        rng = np.random.default_rng(seed)
        P = np.zeros(self.n_params, np.float32)
        S = np.zeros(self.S.numel(), np.float32)
        for p in self.net._all_params:
            kind, scale = p.init
            if kind == 'normal':
                v = (scale * rng.standard_normal(p.size)).astype(np.float32)
                if p.eq is not None:
                    v = v + p.eq.reshape(-1)
            elif kind == 'ones':
                v = np.ones(p.size, np.float32)
            else:
                v = np.zeros(p.size, np.float32)
            (P if p.trainable else S)[p.offset:p.offset + p.size] = v
        self.P.copy_(torch.from_numpy(P))
        self.S.copy_(torch.from_numpy(S))
        self.A.zero_()
        self.invalidate_packs()


    # ------------------------------------------------------------------ buffers
    def _ensure_capacity(self, n, train=True):
        """Device buffers for batches of up to n samples.  The evaluation path ('ev': forward only, any
        batch size -- the statistics pass of scripts/lib/desc.py:10-22 feeds thousands of images per
        launch) allocates only what a forward pass touches; the gradient buffers follow the largest
        TRAINING batch seen."""
        dev = self.dev
        z = lambda *shape: torch.zeros(shape, device=dev)
        if n > self.n_max:
            biggest = max([b.H[i] * b.W[i] * b.C[i] for b in self.blocks for i in range(b.L)] + [int(np.prod(self.x0_shape))])
            if n * biggest >= 2 ** 30:
                raise ValueError('batch of %d samples: the kernels address an activation tensor with 32-bit byte offsets' % n)
            self.n_max = n
            self._progs.clear()
            self._graphs.clear()
            self._gen = getattr(self, '_gen', 0) + 1       # (buffer generation: lib/_co.py rebuilds its merged program)
            self.n_max_bwd = 0
            h, w, c0 = self.x0_shape
            self.x0 = z(n, h, w, c0)
            self.y = z(n, self.n_cls)
            self.k_cpt = z(n)
            for b in self.blocks:
                b.s = [z(n, b.H[i], b.W[i], b.C[i]) for i in range(b.L)]
                b.sp = [z(n, b.H[i] // 2, b.W[i] // 2, b.C[i]) for i in range(b.L - 1)]     # 2x2-max-pooled s
                if b.has_exit:
                    b.z = z(n, self.n_cls) if b.head is not None else None
                    if b.router is not None:
                        R, R2 = (b.router.comps[k].hypers.n_chan for k in (1, 4))
                        b.R, b.R2 = R, R2
                        b.h1, b.h2 = z(n, R), z(n, R2)
                        b.bn_save = z(2 * R + 2 * R2)
            nn, nl, ns = len(self.nodes), len(self.leaves), max(len(self.switches), 1)
            self.p_tr, self.p_ev = z(nn * n), z(nn * n)
            self.w_cerr = z(nl * n)
            self.dr = z(ns * n * self.max_sinks)
            # One allocation that the evaluation path clears with the launch that packs the weights:
            # loss sums | per-block sample counts of the routed evaluation | r | c_err | d_cor
            # (routed evaluation only writes the entries of samples that REACH a node).
            nb = (len(self.blocks) + 3) // 4 * 4
            n_r, n_l = ns * n * self.max_sinks, nl * n
            self._ev_arena = torch.zeros(32 + 4 * nb + 4 * ((n_r + 2 * n_l + 3) // 4 * 4), dtype=torch.uint8, device=dev)
            self.loss_ev = self._ev_arena[:32].view(torch.float64)
            self.ev_cnt = self._ev_arena[32:32 + 4 * nb].view(torch.int32)
            fl = self._ev_arena[32 + 4 * nb:].view(torch.float32)
            self.r, self.c_err, self.d_cor = fl[:n_r], fl[n_r:n_r + n_l], fl[n_r + n_l:n_r + 2 * n_l]
            for k, b in enumerate(self.blocks):
                b.ev_idx = torch.zeros(n, dtype=torch.int32, device=dev)     # samples routed to this block ('ev')
                b.ev_cnt = self.ev_cnt[k:k + 1]
        if train and n > self.n_max_bwd:
            self.n_max_bwd = n
            self._gen = getattr(self, '_gen', 0) + 1
            self._progs = {k: v for k, v in self._progs.items() if k[0] != 'tr'}
            self._graphs = {k: v for k, v in self._graphs.items() if k[0] not in ('tr', 'trK')}
            for b in self.blocks:
                b.dzg = [z(n, b.H[i], b.W[i], b.C[i]) for i in range(b.L)]
                if b.has_exit:
                    b.dx = z(n, b.H[-1] * b.W[-1] * b.C[-1])
                    b.dzh = z(n, self.n_cls) if b.head is not None else None
                    if b.router is not None:
                        b.dh1 = z(n, b.R)
                        b.dh2 = z(n, b.R2) if (self.generic_exits or n > 128) else None     # (scratch of mpnn_exit_tail_bwd_gen)


    # ------------------------------------------------------------------ programs
    @staticmethod
    def _nslot(b, i):
        """Statistics slots of scale i: many workgroups -> many slots; few -> few (every consumer
        workgroup re-adds the slots in its prologue)."""
        return 8          # measured: 8 everywhere beats 16/16/8/4 by 1.6 % (one slot-sum round trip in every consumer)


    def _bn(self, b, i, with_sum=True):
        bn = b.bns[i].params
        return dict(sum=self.dsum[b.sum_off[i]:] if with_sum else None, gamma=bn.γ.data, beta=bn.β.data,
                    m_avg=bn.m_avg.data, v_avg=bn.v_avg.data, eps=float(b.bns[i].hypers.ϵ),
                    nslot=self._nslot(b, i))


    def _act_of_input(self, b, i, n, mode, fwd=False):
        """mpnn_act of the block's input at scale i."""
        if b.in_map is None:
            if fwd and getattr(self, 'rgbx_probe', False) and b.in_shift[i] > 0:
                # TIMING PROBE (tools/rgbx_probe.py; results are only right while x4 holds the strided picks of x0): the
                # pyramid scale as a dense 4-channel map (RGBX, X = 0) -- aligned float4 pixels, no address shift
                return _hip.act(self.x4[b.in_shift[i]], 4, _hip.ACT_IDENTITY, 0)
            return _hip.act(self.x0, self.x0_shape[2], _hip.ACT_IDENTITY, b.in_shift[i])
        pb, j = b.parent, b.in_map[i]
        return _hip.act(pb.s[j], pb.C[j], mode, 0, self._bn(pb, j), n * pb.H[j] * pb.W[j])


    def _bn_ctx(self, b, i, n, with_red=True):
        ctx = _hip.BnCtx()
        ctx.s = b.s[i].data_ptr()
        ctx.bn = _hip.act(None, b.C[i], _hip.ACT_BN_BATCH, 0, self._bn(b, i), n * b.H[i] * b.W[i])
        ctx.red = self.dred[b.sum_off[i]:].data_ptr() if with_red else None
        ctx.red_nslot = self._nslot(b, i)
        self._keep.append(ctx)
        return ctx
