"""The evaluation programs: dense ('ev': every block on every sample, the reference's schedule) and routed (a block runs on
the samples its ancestors' routers sent to it: sample lists written on the device, a dense prefix made routed after the
fact) -- DESIGN.md section 3, "Routed evaluation"."""
import ctypes as C
import os

import numpy as np
import torch

from lib import _hip
from lib.net_types import n_leaves, params_list_rec
from lib._eng_common import (BLOCK_COMPS, CAPTURE_MODE, HEAD_COMPS, OPT_CHUNK, ROUTER_COMPS, BoundInput, _attr, _Block, _kind, _nf,
                             _Node)


class EvalPrograms:

    def _program_ev(self, n, routed):
        """Forward-only program in evaluation mode (BatchNorm moving averages, layer_types.py:237-238;
        hard routing pi_ev, net_types.py:127-131).

        dense : the reference's schedule -- every block on every sample (11 wavefront launches), then
                ONE mpnn_exit_ev launch for all exits and mpnn_route for p_ev / p_tr.
        routed: the reference multiplies 0/1 masks p_ev into the statistics and still evaluates every
                block densely; here a block only runs on the samples its ancestors' routers sent to it.
                Per tree depth: the block's convs gather their inputs through the block's sample list
                (mpnn_conv_fwd_args.idx/cnt: indirection in the tile loader, results land at the
                samples' own rows), then mpnn_exit_ev evaluates head + router on that list and appends
                each sample to the list of the child it is routed to (wave ballot + prefix sum, count
                on the device).  No host sync anywhere; mpnn_route at the end reads the (cleared,
                then sparsely written) r / c_err / d_cor and produces the same p_ev as the dense pass.
        """
        lib, keep = self.lib, self._keep
        net, kind = self.net, self.net._net_kind
        ϕ = net.hypers
        act_mode = _hip.ACT_BN_MOVING
        fwd = []
        # A geometry the group launch has no body for (64+ channels on 16x16 / 32x32 maps: no shipped spec has one): every
        # conv as its own mpnn_msconv_fwd launch.  That entry point takes no sample lists, so a ROUTED pass of such a net
        # runs every conv densely (d0 beyond the deepest block) and is made routed by mpnn_ev_prefix_walk alone.
        singles = not self._groupable()
        if singles and routed:
            routed = 1 + max(self._depths().values())

        def call(fn, what, *args, flops=0.0, tag=''):
            def launch(st):
                _hip.check(fn(*args, st), what)
            launch.what, launch.flops, launch.tag = what, float(flops), tag
            launch.stream, launch.waits, launch.records = 0, (), None
            launch.args = args
            return launch

        depth = {}
        for b in self.blocks:
            depth[id(b)] = 0 if b.parent is None else depth[id(b.parent)] + 1
        # routed = d0 >= 1: the CONVS of blocks with depth < d0 run on every sample (wavefront groups, like the dense
        # program); from depth d0 on a block's convs gather through its sample list.  Every EXIT runs on its block's list
        # (so that r / c_err / d_cor are only written where a sample reaches the node), whatever the depth.
        d0 = int(routed)
        # The dense prefix's EXITS in one launch as well (d0 >= 2): a routed pass is a chain of (conv, exit) launches per
        # depth, each exit gated by the router above it -- d0 serial exit launches for blocks whose convs run on every
        # sample anyway.  Their exits run densely in ONE launch instead; mpnn_ev_prefix_walk then clears the entries of the
        # samples that do not reach a node and writes the lists of the blocks at depth d0 (csrc/exit_ev.hip).  Same results.
        def src_of(b):
            # the nearest switch above block b and the sink of it that leads to b (None: every sample reaches b)
            child, p = b, b.parent
            while p is not None and p.router is None:
                child, p = p, p.parent
            return None if p is None else (p, p.sink_blocks.index(child))
        prefix = [b for b in self.blocks if routed and depth[id(b)] < d0]
        walk = bool(routed) and d0 >= 2 and os.environ.get('MPNN_EV_PREFIX_WALK', '1') != '0' and \
            sum(1 for b in prefix if b.has_exit) <= _hip.PREFIX_MAX and \
            sum(1 for b in self.blocks if depth[id(b)] == d0) <= _hip.PREFIX_MAX
        in_prefix = {id(b) for b in prefix} if walk else set()
        # sample lists: a block below a dynamic switch owns one; below a static node it shares its parent's
        for b in self.blocks:
            par = b.parent
            if not routed or par is None or id(b) in in_prefix:
                b.ev_list = None
            elif walk and depth[id(b)] == d0:          # (frontier: its list comes from the prefix walk)
                b.ev_list = (b.ev_idx, b.ev_cnt) if src_of(b) is not None else None
            elif par.router is not None:
                b.ev_list = (b.ev_idx, b.ev_cnt)
            else:
                b.ev_list = par.ev_list
            b.ev_conv_list = b.ev_list if (routed and depth[id(b)] >= d0) else None

        def fwd_args(b, i, a):
            cp = b.conv.params
            a.a = self._act_of_input(b, i, n, act_mode)
            if i > 0:
                a.v, a.Cv = b.sp[i - 1].data_ptr(), b.C[i - 1]
                a.wv_pack = self.packs[b.pack['w_vert_%i' % (i - 1)][0]:].data_ptr()
            if i < b.L - 1:
                a.pool_out = b.sp[i].data_ptr()
            a.wa_pack = self.packs[b.pack['w_horz_%i' % i][0]:].data_ptr()
            a.bias = getattr(cp, 'b_%i' % i).data.data_ptr()
            a.out = b.s[i].data_ptr()
            a.out_sum = None
            a.out_nslot = self._nslot(b, i)
            a.n, a.H, a.W, a.Cout = n, b.H[i], b.W[i], b.C[i]
            if b.ev_conv_list is not None:
                a.idx, a.cnt = b.ev_conv_list[0].data_ptr(), b.ev_conv_list[1].data_ptr()

        fl_f = lambda b, i: 2.0 * n * b.H[i] * b.W[i] * 9 * b.C[i] * (b.Cin[i] + (b.C[i - 1] if i > 0 else 0))
        tag_f = lambda b, i: 'h%d %d+%d->%d' % (b.H[i], b.Cin[i], b.C[i - 1] if i > 0 else 0, b.C[i])

        def group_launches(members):
            if singles:
                for b, i in members:
                    a = _hip.ConvFwdArgs()
                    fwd_args(b, i, a)
                    keep.append(a)
                    fwd.append(call(lib.mpnn_msconv_fwd, 'fwd', C.byref(a), flops=fl_f(b, i), tag=tag_f(b, i)))
                return
            for c0 in range(0, len(members), 4):
                grp = members[c0:c0 + 4]
                arr = (_hip.ConvFwdArgs * len(grp))()
                for a, (b, i) in zip(arr, grp):
                    fwd_args(b, i, a)
                dev_arr = _hip.to_device_table(list(arr), self.dev)
                keep.extend([arr, dev_arr])
                fwd.append(call(lib.mpnn_msconv_fwd_group, 'fwd_group', arr, dev_arr.data_ptr(), len(grp),
                                flops=sum(fl_f(b, i) for b, i in grp), tag=' | '.join(tag_f(b, i) for b, i in grp)))

        # ---- exit records ----
        dyn = bool(getattr(ϕ, 'dyn_k_cpt', False))
        MS = self.max_sinks
        recs = {}
        for b in self.blocks:
            if not b.has_exit:
                continue
            L1 = b.L - 1
            e = _hip.ExitEvArgs()
            e.a = _hip.act(b.s[L1], b.C[L1], act_mode, 0, self._bn(b, L1, with_sum=False), n * b.H[L1] * b.W[L1])
            e.HW, e.n = b.H[L1] * b.W[L1], n
            if b.head is not None:
                lt, ce = b.head.layer.comps[1], b.head.layer.comps[3]
                leaf = b.head.leaf_id
                e.w_head, e.b_head, e.n_cls = lt.params.w.data.data_ptr(), lt.params.b.data.data_ptr(), self.n_cls
                e.y, e.eps_ce = self.y.data_ptr(), float(ce.hypers.ϵ)
                e.c_err, e.d_cor = self.c_err[leaf * n:].data_ptr(), self.d_cor[leaf * n:].data_ptr()
            if b.router is not None:
                rc = b.router.comps
                l1, bn1, l2, bn2, l3 = rc[1], rc[2], rc[4], rc[5], rc[7]
                sw = b.node.switch_id
                D = lambda prm: prm.data.data_ptr()
                e.w1, e.b1, e.R, e.n_sinks, e.R2 = D(l1.params.w), D(l1.params.b), b.R, len(b.node.layer.sinks), b.R2
                e.extra_col, e.k_cpt, e.alpha_cpt = (1 if dyn else 0), self.k_cpt.data_ptr(), float(_attr(ϕ, 'α_cpt', 0.0))
                e.g1, e.be1, e.m1, e.v1 = D(bn1.params.γ), D(bn1.params.β), D(bn1.params.m_avg), D(bn1.params.v_avg)
                e.w2, e.bias2 = D(l2.params.w), D(l2.params.b)
                e.g2, e.be2, e.m2, e.v2 = D(bn2.params.γ), D(bn2.params.β), D(bn2.params.m_avg), D(bn2.params.v_avg)
                e.w3, e.bias3 = D(l3.params.w), D(l3.params.b)
                e.bn_eps, e.bn_eps2 = float(bn1.hypers.ϵ), float(bn2.hypers.ϵ)
                e.r, e.r_stride = self.r[sw * n * MS:].data_ptr(), MS
                if routed and id(b) not in in_prefix:       # (a prefix exit runs on every sample: the walk writes the lists)
                    for i, sb in enumerate(b.sink_blocks):
                        if sb is not None:
                            e.child_idx[i], e.child_cnt[i] = sb.ev_idx.data_ptr(), sb.ev_cnt.data_ptr()
            if b.ev_list is not None:
                e.idx, e.cnt = b.ev_list[0].data_ptr(), b.ev_list[1].data_ptr()
            if self.generic_exits:                       # (scratch maps of mpnn_exit_ev_gen)
                e.z = b.z.data_ptr() if b.head is not None else None
                e.h1 = b.h1.data_ptr() if b.router is not None else None
            if not self.generic_exits:
                _hip.check(lib.mpnn_exit_ev_check(C.byref(e)), 'exit_ev record')
            recs[id(b)] = e

        kidx = {h: k for k, h in enumerate(sorted({h for b in self.blocks for h in b.H}, reverse=True))}

        def wavefront(blocks):
            levels = {}
            for b in blocks:
                for i in range(b.L):
                    levels.setdefault(depth[id(b)] + kidx[b.H[i]], []).append((b, i))
            for d in sorted(levels):
                group_launches(levels[d])

        def exits_of(blocks):
            order = [recs[id(b)] for b in blocks if id(b) in recs]
            if order:
                tab = _hip.to_device_table(order, self.dev)
                keep.append(tab)
                fwd.append(call(lib.mpnn_exit_ev_gen if self.generic_exits else lib.mpnn_exit_ev, 'exit_ev', tab.data_ptr(), len(order), n))

        if not routed:
            wavefront(self.blocks)
            exits_of(self.blocks)
        else:
            by_depth = {}
            for b in self.blocks:
                by_depth.setdefault(depth[id(b)], []).append(b)
            wavefront([b for b in self.blocks if depth[id(b)] < d0])
            if walk:
                exits_of(prefix)
                pa, rec_of = _hip.EvPrefixArgs(), {}
                pa.n = n
                for b in prefix:
                    if not b.has_exit:
                        continue
                    j = rec_of[id(b)] = len(rec_of)
                    src = src_of(b)
                    pa.parent[j], pa.parent_sink[j] = (-1, 0) if src is None else (rec_of[id(src[0])], src[1])
                    e = recs[id(b)]
                    if b.router is not None:
                        pa.n_sinks[j], pa.r_stride[j], pa.r[j] = e.n_sinks, e.r_stride, e.r
                    if b.head is not None:
                        pa.c_err[j], pa.d_cor[j] = e.c_err, e.d_cor
                pa.count = len(rec_of)
                for b in self.blocks:
                    if depth[id(b)] == d0 and b.ev_list is not None:
                        f = pa.n_front
                        src = src_of(b)
                        pa.front_parent[f], pa.front_sink[f] = rec_of[id(src[0])], src[1]
                        pa.front_idx[f], pa.front_cnt[f] = b.ev_idx.data_ptr(), b.ev_cnt.data_ptr()
                        pa.n_front = f + 1
                if pa.count > 0:                   # (a prefix of static blocks only has no exit to make routed)
                    dev_pa = _hip.to_device_table([pa], self.dev)
                    keep.extend([pa, dev_pa])
                    fwd.append(call(lib.mpnn_ev_prefix_walk, 'ev_prefix_walk', C.byref(pa), dev_pa.data_ptr()))
            for d in sorted(by_depth):
                bs = by_depth[d]
                if walk and d < d0:
                    continue
                if d >= d0:
                    for i in range(max(b.L for b in bs)):
                        members = [(b, i) for b in bs if i < b.L]
                        # a launch holds members that all carry a list, or none (the root block: every sample)
                        for with_list in (False, True):
                            part = [(b, i) for b, i in members if (b.ev_conv_list is not None) == with_list]
                            if part:
                                group_launches(part)
                exits_of(bs)            # (the exits of one depth: their lists come from the depth above)

        ra = self._route_args(n, 'ev', self.loss_ev)
        fwd.append(call(lib.mpnn_route, 'route', C.byref(ra)))
        return dict(fwd=fwd, bwd=[], n=n, mode='ev', routed=routed)
