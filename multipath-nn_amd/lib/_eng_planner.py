"""The training program of a net: the launch lists of one step ('tr') -- forward wavefront groups, exit path, backward
dependency levels with their workgroup budgets, the finishing launch -- built once per (batch, planner settings) and
cached.  `program(mode, n, routed)` is the entry point; evaluation programs: lib/_eng_eval.py."""
import ctypes as C
import os

import numpy as np
import torch

from lib import _hip
from lib.net_types import n_leaves, params_list_rec
from lib._eng_common import (BLOCK_COMPS, CAPTURE_MODE, HEAD_COMPS, OPT_CHUNK, ROUTER_COMPS, BoundInput, _attr, _Block, _kind, _nf,
                             _Node)


class Planner:

    def _wsplit(self, b, i, n, fused=False):
        """Workgroups the pixel range of a wgrad launch is divided over."""
        H = b.H[i]
        tiles = n * (H // 16) * (H // 4) if H >= 16 else (n if H == 8 else (n + 3) // 4)
        nch = (b.Cin[i] + 15) // 16 + ((b.C[i - 1] + 15) // 16 if i > 0 else 0)
        groups = max(1, b.C[i] // 64) if b.C[i] % 64 == 0 else (b.C[i] // 32 if b.C[i] % 32 == 0 else b.C[i] // 16)
        if fused:
            groups = b.C[i] // 64 if b.C[i] % 64 == 0 else b.C[i] // 16    # as mpnn_msconv_bwd_scale
        cap = int(os.environ.get('MPNN_WG_CAP', '512'))
        budget = cap
        if fused:
            # about half of the workgroups that are resident at once: the dgrad bodies of the same
            # launch take the rest, and everything starts together
            has_dgrad = 1 if (b.in_map is not None or i > 0) else 0
            dg_items = tiles * ((b.parent.C[b.in_map[i]] // 16 if b.parent is not None else 0) + (b.C[i - 1] // 16 if i > 0 else 0))
            slots = self.lib.mpnn_msconv_bwd_scale_slots(b.H[i], b.W[i], b.C[i], has_dgrad, 1 if i > 0 else 0, dg_items)
            if slots > 0:
                div = float(os.environ.get('MPNN_WG_DIV', '2'))
                budget = min(cap, int(slots / div) if has_dgrad else slots)      # (a third / a quarter: measured slower)
                if has_dgrad and b.C[i] % 64 == 0 and dg_items > slots // 3:
                    # a 64-channel layer with three workgroups per CU: the input-gradient bodies get one workgroup
                    # per (tile, row) if that fits, the weight gradients the rest
                    budget = min(budget, max(slots - dg_items, slots // 4))
        want = max(1, budget // (nch * groups))
        w_bytes = 4 * 9 * b.C[i] * (b.Cin[i] + (b.C[i - 1] if i > 0 else 0))
        want = min(want, max(1, (12 << 20) // w_bytes))         # keep a layer's slab under ~12 MB
        want = max(1, min(tiles, want))
        return self._xcd_round(want)


    @staticmethod
    def _xcd_round(g):
        """Workgroups per row of an XCD-aware launch (conv_kernel.h, ConvP::xcd): a multiple of 8 from 16 on."""
        return (g // 8) * 8 if (g >= 16 and os.environ.get('MPNN_XCD', '1') != '0') else g


    # ------------------------------------------------------------------ backward schedule
    def _bwd_deps(self, b, i):
        """(block, scale) triples B(.) that must have run before B(b, i) = {dgrad-horz, dgrad-vert, weight gradients
        of g(b, i)}: the coarser scale of the block (its dgrad-vert turns dz(b, i) into g(b, i)), the child blocks'
        launches at this scale (their dgrad-horz writes dz(b, i)), and -- because the dgrad-vert of B(b, i) converts
        dz(b, i-1) into g(b, i-1) IN PLACE -- the child blocks' launches at the finer scale as well."""
        deps = []
        if i < b.L - 1:
            deps.append((b, i + 1))
        for c in b.children:
            for j, src in enumerate(c.in_map):
                if src == i or (src == i - 1 and i > 0 and b.has_dz[i - 1]):
                    deps.append((c, j))
        return deps


    def _bwd_schedule(self, order, n):
        """Launch groups of the backward pass: [[((kb, b, i), budget), ...], ...] in execution order.  Triples of one
        dependency level run as ONE launch (mpnn_msconv_bwd_level) when a kernel variant covers their shapes, they
        write different maps (tree nets: siblings accumulate into one parent map -> consecutive launches) and there
        are at most MPNN_BWD_LEVEL_MAX of them; budget = workgroups of each body (None: a plain mpnn_msconv_bwd_scale
        launch, which sizes itself)."""
        level = {}
        for kb, b, i in order:                               # (a topological order)
            level[(id(b), i)] = 1 + max([level[(id(d), j)] for d, j in self._bwd_deps(b, i)], default=-1)
        by_level = {}
        for m in order:
            by_level.setdefault(level[(id(m[1]), m[2])], []).append(m)
        groups = []
        for d in sorted(by_level):
            pend = list(by_level[d])
            while pend:
                grp, targets, rest = [], set(), []
                for m in pend:
                    kb, b, i = m
                    tgt = (id(b.parent), b.in_map[i]) if b.parent is not None else None
                    if len(grp) < _hip.BWD_LEVEL_MAX and (tgt is None or tgt not in targets):
                        grp.append(m)
                        targets.add(tgt)
                    else:
                        rest.append(m)
                pend = rest
                bud = self._level_budget(grp, n) if (len(grp) > 1 or self.co_share > 1) else None
                if bud is None and self.co_share > 1 and len(grp) > 1:
                    # (the members do not fit slots / co_share together: one table-driven launch each)
                    buds = [self._level_budget([m], n) for m in grp]
                    if any(b is None for b in buds):
                        raise NotImplementedError('co-training %d nets: a backward launch does not fit the resident slots' % self.co_share)
                    groups += [[(m, b[0])] for m, b in zip(grp, buds)]
                elif bud is None and self.co_share > 1:
                    raise NotImplementedError('co-training %d nets: a backward launch does not fit the resident slots' % self.co_share)
                elif bud is None:
                    groups += [[(m, None)] for m in grp]
                else:
                    groups.append(list(zip(grp, bud)))
        return groups


    # Budget model of a level launch: relative latency of one work item of a body (a dgrad unit = a 16-channel chunk
    # of g for one 64-pixel tile and one 16-channel output row; a weight-gradient tile), from the phase traces
    # (profiles/) and a sweep of the step time (tools/knob_sweep.sh): dgrad-vert units carry the max-pool /
    # BatchNorm-backward epilogue, a 16-channel weight-gradient tile is cheaper than a dgrad unit (nine-tap
    # accumulation, lean staging), 64-channel groups have four times its MFMAs.
    _LAT = dict(h=1.0, v=1.4, w1=float(os.environ.get('MPNN_LAT_W1', '0.75')), w4=float(os.environ.get('MPNN_LAT_W4', '2.6')))

    # co-trained groups (throughput-bound launches; swept at K = 8: w4 2.6 -> 2 086 us per joint step, 3.4 -> 2 067, 4.5 -> 2 090)
    _LAT_CO = dict(h=1.0, v=1.4, w1=float(os.environ.get('MPNN_LAT_W1', '0.75')), w4=float(os.environ.get('MPNN_LAT_W4', '3.4')))


    def _level_budget(self, grp, n):
        """Workgroups of every body of a level launch: the assignment that minimises the longest serial chain
        (items per workgroup x item latency) over all bodies with everything resident at once -- small members get
        (nearly) one item per workgroup, the large member the rest.  None: no kernel variant covers the shapes."""
        lib = self.lib
        LAT = self._LAT if self.co_share == 1 else self._LAT_CO
        H = (C.c_int * len(grp))(*[b.H[i] for _, b, i in grp])
        W = (C.c_int * len(grp))(*[b.W[i] for _, b, i in grp])
        Co = (C.c_int * len(grp))(*[b.C[i] for _, b, i in grp])
        slots = lib.mpnn_msconv_bwd_level_slots(H, W, Co, len(grp))
        if slots <= 0:
            return None
        slots //= self.co_share
        bodies = []                                          # (member, kind, rows, tiles, latency per item)
        for k, (kb, b, i) in enumerate(grp):
            tiles = lib.mpnn_wgrad_tiles(n, b.H[i], b.W[i])
            units = b.C[i] // 16
            if b.parent is not None:
                bodies.append((k, 'h', b.parent.C[b.in_map[i]] // 16, tiles, LAT['h'] * units))
            if i > 0:
                bodies.append((k, 'v', b.C[i - 1] // 16, tiles, LAT['v'] * units))
            ot = 4 if b.C[i] % 64 == 0 else 1
            nch = (b.Cin[i] + 15) // 16 + ((b.C[i - 1] + 15) // 16 if i > 0 else 0)
            bodies.append((k, 'w', nch * (b.C[i] // (16 * ot)), tiles, LAT['w%d' % ot]))
        if sum(rows for _, _, rows, _, _ in bodies) > slots:
            return None

        def fit(T):                                          # workgroups per row of each body for a chain of at most T
            gx = []
            for _, _, rows, tiles, lat in bodies:
                per = int(T / lat + 1e-9)
                if per < 1:
                    return None
                gx.append(-(-tiles // per))
            return gx if sum(g * body[2] for g, body in zip(gx, bodies)) <= slots else None
        cands = sorted({lat * j for _, _, _, tiles, lat in bodies for j in range(1, tiles + 1)})
        lo, hi = 0, len(cands) - 1
        while lo < hi:
            mid = (lo + hi) // 2
            if fit(cands[mid]) is not None:
                hi = mid
            else:
                lo = mid + 1
        gx = fit(cands[lo])
        if gx is None:
            return None
        # left-over slots: to the bodies with the longest chain
        used = sum(g * body[2] for g, body in zip(gx, bodies))
        while True:
            cand = [q for q in range(len(gx)) if gx[q] < bodies[q][3] and used + bodies[q][2] <= slots]
            if not cand:
                break
            q = max(cand, key=lambda q: -(-bodies[q][3] // gx[q]) * bodies[q][4])
            gx[q] += 1
            used += bodies[q][2]
        if os.environ.get('MPNN_PLAN_DEBUG'):
            print('level budget: %d slots, chain %.1f; ' % (slots, cands[lo]) + '; '.join(
                'm%d %s rows %d tiles %d -> gx %d (%d wgs, %d items/wg)' % (k, kind, rows, tiles, g, g * rows, -(-tiles // g))
                for (k, kind, rows, tiles, lat), g in zip(bodies, gx)))
        out = [dict(gxh=0, gxv=0, split=1) for _ in grp]
        for (k, kind, rows, tiles, lat), g in zip(bodies, gx):
            if self.co_share == 1:              # (co-trained groups: no XCD-aware order, no rounding of slots / K: see conv_fwd.hip)
                g = self._xcd_round(g)
            if kind == 'w':
                kb, b, i = grp[k]
                w_bytes = 4 * 9 * b.C[i] * (b.Cin[i] + (b.C[i - 1] if i > 0 else 0))
                g = max(1, min(g, max(1, (12 << 20) // w_bytes)))      # keep a layer's slab under ~12 MB
            out[k]['gxh' if kind == 'h' else 'gxv' if kind == 'v' else 'split'] = int(g)
        return out


    def program(self, mode, n, routed=False):
        """Launch lists of one (mode, batch size).  routed ('ev' only): the routed evaluation -- every
        block runs on the sample list its parent's router produced on the device (see _program_ev)."""
        try:
            return self._program(mode, n, routed)
        finally:
            self.lib.mpnn_set_reserved_cus(0)      # (a data-parallel training program is built with a reservation in place)


    def routed_prefix(self, n):
        """Depth from which the routed evaluation gathers (>= 1; see _program_ev).  The blocks above it run on every
        sample in wavefront-grouped launches: early blocks lose few samples, so routing them saves little work and
        costs the block-serial schedule (one launch per scale, then the exit, per block) -- which is what made the
        fully routed program slower than the dense one below ~2 000 samples (profiles/r04_eval_sweep.txt).
        MPNN_ROUTED_PREFIX overrides the batch-size rule."""
        env = os.environ.get('MPNN_ROUTED_PREFIX')
        if env:
            return max(1, int(env))
        for lim, d0 in self._ROUTED_PREFIX:
            if n >= lim:
                return d0
        return self._ROUTED_PREFIX[-1][1]

    _ROUTED_PREFIX = ((6144, 1), (3072, 2), (1536, 3), (640, 4), (0, 6))     # (re-swept with the prefix walk: profiles/r05_eval_prefix_sweep.txt)


    def _program(self, mode, n, routed):
        # routed='auto': routed above ROUTED_MIN_BATCH samples, dense below (the routed schedule is block-serial --
        # 28 launches against 13 -- and only pays once the launches are throughput-bound; profiles/r03_eval_sweep.txt)
        if routed == 'auto':
            routed = n >= self.routed_min_batch
        explicit = routed if (isinstance(routed, int) and not isinstance(routed, bool) and routed >= 1) else None
        routed = bool(routed) and mode != 'tr' and bool(self.switches) and self.net._net_kind != 'sr'
        if routed:                                   # (an int >= 1: blocks of a smaller depth run on every sample)
            routed = explicit if explicit is not None else self.routed_prefix(n)
        if mode == 'tr' and self.allreduce is not None and self.multi_stream:
            # one section would fork and re-join the same side streams twice inside one capture (ROCm 7.2 crashes in
            # hipStreamEndCapture), and the DAG schedule has no bucket boundaries to overlap the collectives with
            raise NotImplementedError('data-parallel training runs on the single-stream schedule (MPNN_STREAMS=0)')
        dp = mode == 'tr' and self.allreduce is not None
        reserve = self.dp_reserve_cus if (dp and len(self.dp_buckets) > 1) else 0
        key = (mode, n, self.multi_stream, self.group_fwd, routed, dp, self.bwd_levels, self.fold_clear, reserve, self.fuse_opt, self.co_share)
        if key in self._progs:
            return self._progs[key]
        self._ensure_capacity(n, mode == 'tr')
        if mode != 'tr':
            prog = self._progs[key] = self._program_ev(n, routed)
            return prog
        lib, keep = self.lib, self._keep
        if self.multi_stream and any(len(b.children) > 1 for b in self.blocks):
            raise NotImplementedError('the multi-stream schedule serialises nothing between sibling blocks that '
                                      'accumulate into one gradient map: tree nets run on the single-stream schedule')
        act_mode = _hip.ACT_BN_BATCH if mode == 'tr' else _hip.ACT_BN_MOVING
        net, kind = self.net, self.net._net_kind
        ϕ = net.hypers
        fwd, bwd = [], []

        cur_reserve = [0]                     # compute units the launches being built leave free (see below: trunk backward)

        def call(fn, what, *args, flops=0.0, tag='', stream=0, waits=(), records=None, host=None):
            def launch(st):
                _hip.check(fn(*args, st), what)
            launch.what, launch.flops, launch.tag = what, float(flops), tag
            launch.stream, launch.waits, launch.records = stream, tuple(waits), records
            launch.args, launch.fn, launch.host = args, fn, host        # (host: the launch's records in host memory, for lib/_co.py)
            launch.reserve = cur_reserve[0]
            return launch

        def marker(kind, tag=''):             # 'fork' / 'join' of the side streams; 'bucket': a gradient range is final
            def launch(st):
                pass
            launch.what, launch.flops, launch.tag = kind, 0.0, tag
            launch.stream, launch.waits, launch.records = 0, (), None
            return launch

        # Streams: 0 = main (the 4x4 maps: the critical path through every block); 1.. = one per
        # larger map size; the last two = weight-gradient side streams (leaves of the DAG).
        sizes = sorted({h for b in self.blocks for h in b.H}, reverse=True)
        sid = {h: (0 if h == sizes[-1] else 1 + k) for k, h in enumerate(sizes)}
        n_scale_streams = len(sizes)
        wg_streams = (n_scale_streams, n_scale_streams + 1)
        self.n_streams = n_scale_streams + 2
        bid = {id(b): k for k, b in enumerate(self.blocks)}
        F = lambda b, i: 'F%d_%d' % (bid[id(b)], i)
        Gn = lambda b, i: 'G%d_%d' % (bid[id(b)], i)

        # ---- forward convs ----
        fwd.append(marker('fork'))

        def fwd_args(b, i, a):
            cp = b.conv.params
            a.a = self._act_of_input(b, i, n, act_mode, fwd=True)
            if i > 0:
                a.v, a.Cv = b.sp[i - 1].data_ptr(), b.C[i - 1]
                a.wv_pack = self.packs[b.pack['w_vert_%i' % (i - 1)][0]:].data_ptr()
            if i < b.L - 1:
                a.pool_out = b.sp[i].data_ptr()
            a.wa_pack = self.packs[b.pack['w_horz_%i' % i][0]:].data_ptr()
            a.bias = getattr(cp, 'b_%i' % i).data.data_ptr()
            a.out = b.s[i].data_ptr()
            a.out_sum = self.dsum[b.sum_off[i]:].data_ptr() if mode == 'tr' else None
            a.out_nslot = self._nslot(b, i)
            a.n, a.H, a.W, a.Cout = n, b.H[i], b.W[i], b.C[i]

        fl_f = lambda b, i: 2.0 * n * b.H[i] * b.W[i] * 9 * b.C[i] * (b.Cin[i] + (b.C[i - 1] if i > 0 else 0))
        tag_f = lambda b, i: 'h%d %d+%d->%d' % (b.H[i], b.Cin[i], b.C[i - 1] if i > 0 else 0, b.C[i])
        groupable = all(b.parent is not None or b.in_map is None for b in self.blocks) and \
            all(c % 16 == 0 and not (c % 64 == 0 and h >= 16) for b in self.blocks for c, h in zip(b.C, b.H))
        if self.group_fwd and not self.multi_stream and groupable and all(len(self.nodes[b.node.parent].layer.sinks) >= 1 for b in self.blocks):
            # Wavefront over the block x scale grid: F(b, k) needs only F(b-1, k) and F(b, k-1), so level
            # d = depth(b) + k is one launch of mutually independent convs.
            kidx = {h: k for k, h in enumerate(sizes)}
            depth = {}
            for b in self.blocks:
                depth[id(b)] = 0 if b.parent is None else depth[id(b.parent)] + 1
            levels = {}
            for b in self.blocks:
                for i in range(b.L):
                    levels.setdefault(depth[id(b)] + kidx[b.H[i]], []).append((b, i))
            for d in sorted(levels):
                members = levels[d]
                for c0 in range(0, len(members), 4):
                    grp = members[c0:c0 + 4]
                    arr = (_hip.ConvFwdArgs * len(grp))()
                    for a, (b, i) in zip(arr, grp):
                        fwd_args(b, i, a)
                    dev_arr = _hip.to_device_table(list(arr), self.dev)
                    keep += [arr, dev_arr]
                    if self.co_share > 1:
                        # (one net of a co-trained group stepping by itself: the grids it has inside the joint launches)
                        fwd.append(call(lib.mpnn_msconv_fwd_group_rep, 'fwd_group', arr, dev_arr.data_ptr(), len(grp), 1, self.co_share,
                                        flops=sum(fl_f(b, i) for b, i in grp), tag=' | '.join(tag_f(b, i) for b, i in grp)))
                        continue
                    fwd.append(call(lib.mpnn_msconv_fwd_group, 'fwd_group', arr, dev_arr.data_ptr(), len(grp),
                                    flops=sum(fl_f(b, i) for b, i in grp),
                                    tag=' | '.join(tag_f(b, i) for b, i in grp)))
        else:
            for b in self.blocks:
                for i in range(b.L):
                    a = _hip.ConvFwdArgs()
                    fwd_args(b, i, a)
                    keep.append(a)
                    fwd.append(call(lib.mpnn_msconv_fwd, 'msconv_fwd', C.byref(a), flops=fl_f(b, i), tag=tag_f(b, i),
                                    stream=sid[b.H[i]], waits=[F(b, i - 1)] if i > 0 else [], records=F(b, i)))
        fwd.append(marker('join'))

        # ---- exits ----
        dyn = bool(getattr(ϕ, 'dyn_k_cpt', False))
        lin_f, lin_b, tail_f, tail_b = [], [], [], []
        MS = self.max_sinks
        kmax = 0
        for b in self.blocks:
            if not b.has_exit:
                continue
            L1 = b.L - 1
            K = b.H[L1] * b.W[L1] * b.C[L1]
            kmax = max(kmax, K)
            lf, lb = _hip.LinFwdArgs(), _hip.LinBwdArgs()
            tf, tb = _hip.ExitTailArgs(), _hip.ExitTailBwdArgs()
            a_in = _hip.act(b.s[L1], b.C[L1], act_mode, 0, self._bn(b, L1), n * b.H[L1] * b.W[L1])
            lf.a, lb.a = a_in, a_in
            lf.HW = lb.HW = b.H[L1] * b.W[L1]
            lf.n = lb.n = tf.n = n
            lf.k_cpt = lb.k_cpt = self.k_cpt.data_ptr()
            lf.alpha_cpt = lb.alpha_cpt = float(_attr(ϕ, 'α_cpt', 0.0))
            if K >= 512 and n <= 512:
                # K-slices for mpnn_lin_fwd (one workgroup per 16 rows pulled all of W through one compute
                # unit); small batches only -- with thousands of rows the launch has workgroups enough
                rg = (n + 15) // 16
                kpart = torch.empty(rg * _hip.LIN_KSLICES * 512, device=self.dev)
                kcnt = torch.zeros(rg, dtype=torch.int32, device=self.dev)
                keep += [kpart, kcnt]
                lf.kpart, lf.kcnt = kpart.data_ptr(), kcnt.data_ptr()
            if mode == 'tr' and n <= 512:
                # row split for mpnn_lin_bwd_rs: partial dW / db tiles of the row groups of a feature block
                nblk = (K + 1 + 63) // 64
                bpart = torch.empty(nblk * _hip.LIN_RSPLIT * _hip.LIN_RS_TILE, device=self.dev)
                bcnt = torch.zeros(nblk, dtype=torch.int32, device=self.dev)
                keep += [bpart, bcnt]
                lb.kpart, lb.kcnt = bpart.data_ptr(), bcnt.data_ptr()
            lb.dx = b.dx.data_ptr()
            if mode == 'tr' and not b.children and not self.multi_stream and not self.generic_exits:
                # the exit's dX is the only gradient of this map: lin_bwd masks it and accumulates the
                # BatchNorm-backward reductions itself (no mpnn_bn_bwd_reduce launch)
                lb.dx = None
                lb.dz_out = b.dzg[L1].data_ptr()
                lb.red_out = self.dred[b.sum_off[L1]:].data_ptr()
                lb.red_nslot = self._nslot(b, L1)
            tf.mode = act_mode
            if b.head is not None:
                lt, ce = b.head.layer.comps[1], b.head.layer.comps[3]
                lf.w[0], lf.b[0], lf.y[0], lf.M[0] = lt.params.w.data.data_ptr(), lt.params.b.data.data_ptr(), b.z.data_ptr(), self.n_cls
                lb.w[0], lb.dy[0], lb.M[0] = lf.w[0], b.dzh.data_ptr(), self.n_cls
                lb.dw[0], lb.db[0] = lt.params.w.grad.data_ptr(), lt.params.b.grad.data_ptr()
                leaf = b.head.leaf_id
                tf.z, tf.y, tf.n_cls, tf.eps_ce = b.z.data_ptr(), self.y.data_ptr(), self.n_cls, float(ce.hypers.ϵ)
                tf.c_err = self.c_err[leaf * n:].data_ptr()
                tf.d_cor = self.d_cor[leaf * n:].data_ptr()
                tb.w_cerr = self.w_cerr[leaf * n:].data_ptr()
                tb.dz = b.dzh.data_ptr()
            if b.router is not None:
                rc = b.router.comps
                l1, bn1, l2, bn2, l3 = rc[1], rc[2], rc[4], rc[5], rc[7]
                R, S = b.R, len(b.node.layer.sinks)
                sw = b.node.switch_id
                lf.w[1], lf.b[1], lf.y[1], lf.M[1] = l1.params.w.data.data_ptr(), l1.params.b.data.data_ptr(), b.h1.data_ptr(), R
                lb.w[1], lb.dy[1], lb.M[1] = lf.w[1], b.dh1.data_ptr(), R
                lb.dw[1], lb.db[1] = l1.params.w.grad.data_ptr(), l1.params.b.grad.data_ptr()
                lf.extra_col[1] = lb.extra_col[1] = 1 if dyn else 0
                tf.h1, tf.R, tf.n_sinks, tf.R2 = b.h1.data_ptr(), R, S, b.R2
                tf.g1, tf.b1 = bn1.params.γ.data.data_ptr(), bn1.params.β.data.data_ptr()
                tf.m1, tf.v1 = bn1.params.m_avg.data.data_ptr(), bn1.params.v_avg.data.data_ptr()
                tf.w2, tf.bias2 = l2.params.w.data.data_ptr(), l2.params.b.data.data_ptr()
                tf.g2, tf.b2 = bn2.params.γ.data.data_ptr(), bn2.params.β.data.data_ptr()
                tf.m2, tf.v2 = bn2.params.m_avg.data.data_ptr(), bn2.params.v_avg.data.data_ptr()
                tf.w3, tf.bias3 = l3.params.w.data.data_ptr(), l3.params.b.data.data_ptr()
                tf.h2 = b.h2.data_ptr()
                tf.r, tf.r_stride = self.r[sw * n * MS:].data_ptr(), MS
                tf.bn_save = b.bn_save.data_ptr()
                tf.bn_eps, tf.bn_decay = float(bn1.hypers.ϵ), float(bn1.hypers.d)
                tf.bn_eps2, tf.bn_decay2 = float(bn2.hypers.ϵ), float(bn2.hypers.d)
                tb.dr = self.dr[sw * n * MS:].data_ptr()
                tb.dh1 = b.dh1.data_ptr()
                if mode == 'tr' and getattr(b, 'dh2', None) is not None:
                    tb.dh2 = b.dh2.data_ptr()
                tb.dg1, tb.db1 = bn1.params.γ.grad.data_ptr(), bn1.params.β.grad.data_ptr()
                tb.dw2, tb.dbias2 = l2.params.w.grad.data_ptr(), l2.params.b.grad.data_ptr()
                tb.dg2, tb.db2 = bn2.params.γ.grad.data_ptr(), bn2.params.β.grad.data_ptr()
                tb.dw3, tb.dbias3 = l3.params.w.grad.data_ptr(), l3.params.b.grad.data_ptr()
            tb.f = tf
            lin_f.append(lf); lin_b.append(lb); tail_f.append(tf); tail_b.append(tb)
        n_exit = len(lin_f)
        # A training step without a clearing launch: the slot sums are cleared by their last reader (the launch that
        # ends the backward pass), the accumulators of mpnn_route by the launch before it (see run()).
        fold = mode == 'tr' and n_exit > 0 and self.fold_clear
        if fold:
            tail_f[0].clear_f, tail_f[0].n_clear_f = self.node_stat.data_ptr(), self.node_stat.numel()
            tail_f[0].clear_d, tail_f[0].n_clear_d = self.loss.data_ptr(), self.loss.numel()
        t_lf, t_lb = _hip.to_device_table(lin_f, self.dev), _hip.to_device_table(lin_b, self.dev)
        t_tf, t_tb = _hip.to_device_table(tail_f, self.dev), _hip.to_device_table(tail_b, self.dev)
        keep += [t_lf, t_lb, t_tf, t_tb]
        if n_exit and self.generic_exits:
            fwd.append(call(lib.mpnn_lin_fwd_gen, 'lin_fwd', t_lf.data_ptr(), n_exit, n, host=lin_f))
            fwd.append(call(lib.mpnn_exit_tail_fwd_gen, 'exit_tail_fwd', t_tf.data_ptr(), n_exit, n, host=tail_f))
        elif n_exit:
            if n <= 512:
                fwd.append(call(lib.mpnn_lin_fwd_ks, 'lin_fwd', t_lf.data_ptr(), n_exit, n, kmax, host=lin_f))
            else:
                fwd.append(call(lib.mpnn_lin_fwd, 'lin_fwd', t_lf.data_ptr(), n_exit, n, host=lin_f))
            # batches beyond the 128 samples the LDS-resident tails hold: the any-width tails (csrc/exit_gen.hip: every pass on
            # 1 024 threads) instead of the tuned kernels' any-size forms -- same records; measured at 256 / 512 / 1 024
            # samples: profiles/r05_train_sweep.txt
            big_tails = n > 128 and bool(int(os.environ.get('MPNN_BIG_TAILS_GEN', '1')))
            fwd.append(call(lib.mpnn_exit_tail_fwd_gen if big_tails else lib.mpnn_exit_tail_fwd, 'exit_tail_fwd', t_tf.data_ptr(), n_exit, n, host=tail_f))

        # ---- route ----
        ra = self._route_args(n, mode, self.loss)
        fwd.append(call(lib.mpnn_route, 'route', C.byref(ra), host=ra))

        prog = dict(fwd=fwd, bwd=bwd, n=n, mode=mode, fold=fold)
        self._progs[key] = prog
        if mode != 'tr':
            return prog

        # ---- backward ----
        slab_plan = dict(size=0)
        level_fix = []
        if n_exit and self.generic_exits:
            bwd.append(call(lib.mpnn_exit_tail_bwd_gen, 'exit_tail_bwd', t_tb.data_ptr(), n_exit, n))
            bwd.append(call(lib.mpnn_lin_bwd_gen, 'lin_bwd', t_lb.data_ptr(), n_exit, n, kmax, host=lin_b))
        elif n_exit:
            bwd.append(call(lib.mpnn_exit_tail_bwd_gen if big_tails else lib.mpnn_exit_tail_bwd, 'exit_tail_bwd', t_tb.data_ptr(), n_exit, n, host=tail_b))
            bwd.append(call(lib.mpnn_lin_bwd_rs if n <= 512 else lib.mpnn_lin_bwd, 'lin_bwd', t_lb.data_ptr(), n_exit, n, kmax, host=lin_b))
        if dp and 'exit' in self.dp_buckets:
            bwd.append(marker('bucket', 'exit'))       # head + router gradients are final: their all-reduce starts here
        # From here to the end of the backward pass a bucket's all-reduce runs beside the launches: their persistent
        # grids (and the workgroup budgets computed below) leave `reserve` compute units to the collective's kernels.
        cur_reserve[0] = reserve
        lib.mpnn_set_reserved_cus(reserve)           # (program() resets it)
        bwd.append(marker('fork'))
        dz_written = set()
        slab_members = []                                   # (is_cut_block, table rows, [(args, field, offset)], optimizer rows)
        slab_params = set()                                 # parameters whose gradient comes out of a slab reduction
        use_levels = self.bwd_levels and not self.multi_stream
        cut_kb = self.dp_cut_block if dp else None

        def make_block(kb, b):
            """Argument builders of one block's backward launches (bound to THIS block)."""
            cp = b.conv.params
            L1 = b.L - 1
            pre = []
            # coarsest scale without a child block: its dy is the exit's dX alone
            if not b.children and (self.multi_stream or not b.has_exit or self.generic_exits):
                ctx = self._bn_ctx(b, L1, n, with_red=False)
                pre.append(call(lib.mpnn_bn_bwd_reduce, 'bn_bwd_reduce', b.dx.data_ptr(), C.byref(ctx),
                                b.dzg[L1].data_ptr(), self.dred[b.sum_off[L1]:].data_ptr(),
                                n * b.H[L1] * b.W[L1], stream=sid[b.H[L1]]))
            # g of the coarsest scale = BatchNorm backward of dz: its own launch in the multi-stream
            # schedule, applied while loading by the three consumers in the fused schedule.
            g_ctx = None
            if self.multi_stream:
                ctx = self._bn_ctx(b, L1, n)
                pre.append(call(lib.mpnn_bn_bwd_apply, 'bn_bwd_apply', b.dzg[L1].data_ptr(), C.byref(ctx),
                                n * b.H[L1] * b.W[L1], stream=sid[b.H[L1]], records=Gn(b, L1)))
            else:
                g_ctx = C.pointer(self._bn_ctx(b, L1, n))

            def vert_args(i):
                a = _hip.DgradVertArgs()
                fine = self._bn_ctx(b, i - 1, n)
                a.g, a.Cg = b.dzg[i].data_ptr(), b.C[i]
                if i == L1 and g_ctx is not None:
                    a.g_ctx = g_ctx
                a.w_pack = self.packs[b.pack['w_vert_%i' % (i - 1)][1]:].data_ptr()
                a.fine = C.pointer(fine)
                a.fine_has_dz = 1 if b.has_dz[i - 1] else 0
                a.dz_g_fine = b.dzg[i - 1].data_ptr()
                a.n, a.H, a.W, a.Cout = n, b.H[i], b.W[i], b.C[i - 1]
                keep.append(a)
                return a

            def horz_args(i):
                pb, j = b.parent, b.in_map[i]
                a = _hip.DgradHorzArgs()
                a.g, a.Cg = b.dzg[i].data_ptr(), b.C[i]
                if i == L1 and g_ctx is not None:
                    a.g_ctx = g_ctx
                a.w_pack = self.packs[b.pack['w_horz_%i' % i][1]:].data_ptr()
                # a map that feeds several child blocks (tree nets): the first child to run writes it
                # (with the exit's dX), the others add their masked share
                first = (id(pb), j) not in dz_written
                dz_written.add((id(pb), j))
                a.accumulate = 0 if first else 1
                a.dy_extra = pb.dx.data_ptr() if (first and pb.has_exit and j == pb.L - 1) else None
                prev = self._bn_ctx(pb, j, n, with_red=False)
                a.prev = C.pointer(prev)
                a.out = pb.dzg[j].data_ptr()
                a.red_out = self.dred[pb.sum_off[j]:].data_ptr()
                a.n, a.H, a.W, a.Cout = n, b.H[i], b.W[i], pb.C[j]
                keep.append(a)
                return a

            def wgrad_args(i, split=None):
                a = _hip.WgradArgs()
                a.a = self._act_of_input(b, i, n, act_mode)
                pa = getattr(cp, 'w_horz_%i' % i)
                pv = getattr(cp, 'w_vert_%i' % (i - 1)) if i > 0 else None
                pb = getattr(cp, 'b_%i' % i)
                if i > 0:
                    a.v, a.Cv = b.sp[i - 1].data_ptr(), b.C[i - 1]
                a.g = b.dzg[i].data_ptr()
                if i == L1 and g_ctx is not None:
                    a.g_ctx = g_ctx
                a.n, a.H, a.W, a.Cout = n, b.H[i], b.W[i], b.C[i]
                if split is None:
                    split = self._wsplit(b, i, n, fused=not self.multi_stream)
                a.n_split = split
                if split == 1:
                    a.dwa, a.db = pa.grad.data_ptr(), pb.grad.data_ptr()
                    a.dwv = pv.grad.data_ptr() if pv is not None else None
                    a.split_stride = 0
                else:
                    sizes = [pa.size, pv.size if pv is not None else 0, pb.size]
                    stride = (sum(sizes) + 3) // 4 * 4
                    off = slab_plan['size']
                    slab_plan['size'] += split * stride
                    rows, ptrs, srows = [], [], []
                    for prm, sz in zip((pa, pv, pb), sizes):
                        if prm is None:
                            continue
                        item = _hip.slab_item_size(split)
                        l2b, eqo, pk = self._opt_info[id(prm)]
                        if pk[1] and pk[1] % 4 == 0:
                            # a weight tensor [9 * Cin][Cout]: items of whole 4-row groups, so that the update applied by
                            # the item's workgroup (mpnn_backward_finish_opt) can write the weight packs as contiguous runs
                            item = min(_hip.SLAB_ITEM, max(item, 4 * pk[2]))
                        for k in range(0, sz, item):
                            cnt = min(item, sz - k)
                            rows += [off + k, prm.offset + k, cnt, split, stride, 0]
                            srows += [prm.offset + k, cnt, prm.node, prm.is_router, l2b, eqo + k if eqo >= 0 else -1,
                                      pk[0], pk[1], pk[2], pk[3], pk[4], 0]
                        slab_params.add(id(prm))
                        ptrs.append((a, {id(pa): 'dwa', id(pb): 'db'}.get(id(prm), 'dwv'), off))
                        off += sz
                    slab_members.append((cut_kb is not None and kb <= cut_kb, rows, ptrs, srows))
                    a.split_stride = stride
                keep.append(a)
                return a

            return pre, vert_args, horz_args, wgrad_args

        fl_v = lambda b, i: 2.0 * n * b.H[i] * b.W[i] * 9 * b.C[i] * b.C[i - 1]
        fl_h = lambda b, i: 2.0 * n * b.H[i] * b.W[i] * 9 * b.C[i] * b.parent.C[b.in_map[i]]
        fl_w = lambda b, i: 2.0 * n * b.H[i] * b.W[i] * 9 * b.C[i] * (b.Cin[i] + (b.C[i - 1] if i > 0 else 0))
        tag_b = lambda b, i: 'h%d %d+%d->%d' % (b.H[i], b.Cin[i], b.C[i - 1] if i > 0 else 0, b.C[i])
        mid_pos = None                                      # index in bwd of the 'mid' slab reduction (filled in below)
        if not self.multi_stream:
            # One launch per (block, scale) -- dgrad-horz, dgrad-vert (which produces g(b,i-1)) and the weight
            # gradients of g(b,i) -- or, with use_levels, one launch per DEPENDENCY LEVEL of those triples
            # (_bwd_schedule): the reversed block order with scales coarsest first is a topological order.
            order = [(kb, b, i) for kb, b in enumerate(reversed(self.blocks)) for i in range(b.L - 1, -1, -1)]
            groups = self._bwd_schedule(order, n) if use_levels else [[(m, None)] for m in order]
            fns = {kb: make_block(kb, b) for kb, b in enumerate(reversed(self.blocks))}
            started = set()
            last_cut = max([g for g, grp in enumerate(groups) for (kb, b, i), _ in grp if cut_kb is not None and kb <= cut_kb],
                           default=None)
            for g, grp in enumerate(groups):
                for (kb, b, i), _ in grp:
                    if kb not in started:
                        started.add(kb)
                        bwd.extend(fns[kb][0])
                built = []
                for (kb, b, i), bud in grp:
                    pre, vert_args, horz_args, wgrad_args = fns[kb]
                    h = horz_args(i) if b.parent is not None else None
                    v = vert_args(i) if i > 0 else None
                    w = wgrad_args(i, None if bud is None else bud['split'])
                    fl = fl_w(b, i) + (fl_h(b, i) if h is not None else 0) + (fl_v(b, i) if v is not None else 0)
                    built.append((h, v, w, bud, fl, tag_b(b, i)))
                if len(built) == 1 and built[0][3] is None:
                    h, v, w, _, fl, tag = built[0]
                    bwd.append(call(lib.mpnn_msconv_bwd_scale, 'bwd_scale',
                                    C.byref(h) if h is not None else None, C.byref(v) if v is not None else None,
                                    C.byref(w), flops=fl, tag=tag))
                else:
                    mem = (_hip.BwdMember * len(built))()
                    for m, (h, v, w, bud, fl, tag) in zip(mem, built):
                        m.horz = C.pointer(h) if h is not None else None
                        m.vert = C.pointer(v) if v is not None else None
                        m.wgrad = C.pointer(w)
                        m.wg_horz, m.wg_vert = bud['gxh'], bud['gxv']
                    rec_bytes = lib.mpnn_msconv_bwd_level_record_size()
                    host = (C.c_char * (rec_bytes * len(built)))()
                    keep.append(mem)
                    # (the slab pointers inside the wgrad records are only known once every slab is laid out:
                    # the records are prepared and uploaded after the loop)
                    level_fix.append((mem, len(built), host, rec_bytes))
                    dev_rec = torch.empty(rec_bytes * len(built), dtype=torch.uint8, device=self.dev)
                    keep.append(dev_rec)
                    level_fix[-1] += (dev_rec,)
                    if self.co_share > 1:      # (one net of a co-trained group by itself: the group's launch form, one copy)
                        bwd.append(call(lib.mpnn_msconv_bwd_level_rep, 'bwd_scale', mem, len(built), 1, dev_rec.data_ptr(),
                                        flops=sum(x[4] for x in built), tag=' | '.join(x[5] for x in built)))
                    else:
                        bwd.append(call(lib.mpnn_msconv_bwd_level, 'bwd_scale', mem, len(built), dev_rec.data_ptr(),
                                        flops=sum(x[4] for x in built), tag=' | '.join(x[5] for x in built)))
                if last_cut is not None and g == last_cut:
                    if any(m[0] for m in slab_members):
                        mid_pos = len(bwd)
                        bwd.append(None)                      # mpnn_slab_reduce of the cut blocks' items (filled in below)
                    bwd.append(marker('bucket', 'mid'))
        else:
            for kb, b in enumerate(reversed(self.blocks)):
                pre, vert_args, horz_args, wgrad_args = make_block(kb, b)
                bwd.extend(pre)
                L1 = b.L - 1
                for i in range(L1, 0, -1):
                    bwd.append(call(lib.mpnn_msconv_dgrad_vert, 'dgrad_vert', C.byref(vert_args(i)), flops=fl_v(b, i),
                                    tag='h%d %d->%d' % (b.H[i], b.C[i], b.C[i - 1]),
                                    stream=sid[b.H[i - 1]], waits=[Gn(b, i)], records=Gn(b, i - 1)))
                if b.parent is not None:
                    for i in range(b.L):
                        bwd.append(call(lib.mpnn_msconv_dgrad_horz, 'dgrad_horz', C.byref(horz_args(i)), flops=fl_h(b, i),
                                        tag='h%d %d->%d' % (b.H[i], b.C[i], b.parent.C[b.in_map[i]]),
                                        stream=sid[b.H[i]]))
                for i in range(b.L):
                    bwd.append(call(lib.mpnn_msconv_wgrad, 'wgrad', C.byref(wgrad_args(i)), flops=fl_w(b, i),
                                    tag=tag_b(b, i), stream=wg_streams[i % 2], waits=[Gn(b, i)]))
        bwd.append(marker('join'))
        keep_ptr = self.dsum_last.data_ptr() if fold else None
        if slab_plan['size']:
            slab = torch.empty(slab_plan['size'], device=self.dev)
            rows, srows, first = [], [], 0
            for want_cut in (True, False):                  # the cut blocks' items first: the 'mid' reduction takes a prefix
                for is_cut, r, ptrs, sr in slab_members:
                    if is_cut == want_cut:
                        rows += r
                        srows += sr
                        for a, field, off in ptrs:
                            setattr(a, field, slab[off:].data_ptr())
                if want_cut:
                    first = len(rows) // 6
            tab = torch.tensor(rows, dtype=torch.int32, device=self.dev)
            keep += [slab, tab]
            n_items = len(rows) // 6
            if mid_pos is not None:
                # data parallel: the conv gradients of the blocks the backward finished first are reduced
                # from their slabs at the bucket boundary (their all-reduce then overlaps the rest of the
                # backward pass); the launch that ends the backward takes the remaining items
                bwd[mid_pos] = call(lib.mpnn_slab_reduce, 'slab_reduce', slab.data_ptr(), self.G.data_ptr(),
                                    tab.data_ptr(), first)
            else:
                first = 0
            if not dp and self.fuse_opt and not self.multi_stream:
                # single process: slab reduction + BatchNorm finalisation + the TALR / momentum update of EVERY parameter
                # as one launch -- each workgroup updates the elements whose gradient it has just produced; the
                # parameters whose gradients were final before (exits; tensors written without slabs) get workgroups
                # of their own
                bn_opt, fused_bn = [], set()
                for b in self.blocks:
                    for i in range(b.L):
                        bn = b.bns[i].params
                        bn_opt += [b.node.idx, int(np.float32(bn.γ.l2).view(np.int32)), int(np.float32(bn.β.l2).view(np.int32)), 0]
                        fused_bn |= {id(bn.γ), id(bn.β)}
                segs = self.seg.cpu().numpy().reshape(-1, _hip.SEG_INTS)
                plain = [segs[k] for k, pid in enumerate(self._seg_owner) if pid not in slab_params and pid not in fused_bn]
                t_seg = torch.tensor(srows, dtype=torch.int32, device=self.dev)
                t_bno = torch.tensor(bn_opt, dtype=torch.int32, device=self.dev)
                t_plain = torch.from_numpy(np.concatenate(plain) if plain else np.zeros(_hip.SEG_INTS, np.int32)).to(self.dev)
                keep += [t_seg, t_bno, t_plain]
                talr = 1 if (self.net._net_kind != 'sr' and getattr(self.net.hypers, 'talr', False)) else 0
                fin = _hip.FinishNet()
                fin.slabs, fin.slab_table, fin.n_items, fin.item_seg = slab.data_ptr(), tab.data_ptr(), n_items, t_seg.data_ptr()
                fin.sums, fin.reds, fin.state = self.dsum.data_ptr(), self.dred.data_ptr(), self.S.data_ptr()
                fin.bn_table, fin.n_bn, fin.bn_opt, fin.n_img, fin.sums_keep = self.bn_table.data_ptr(), self.n_bn, t_bno.data_ptr(), n, keep_ptr
                fin.params, fin.accum, fin.grads = self.P.data_ptr(), self.A.data_ptr(), self.G.data_ptr()
                fin.node_stat, fin.hyp, fin.talr, fin.inv_n, fin.grad_scale = self.node_stat.data_ptr(), self.hyp.data_ptr(), talr, 1.0 / n, 1.0
                fin.w_eq, fin.packs = (self.w_eq.data_ptr() if self.w_eq is not None else None), self.packs.data_ptr()
                fin.plain_seg, fin.n_plain = t_plain.data_ptr(), len(plain)
                prog['finish_net'] = fin                   # (the same arguments as one record: lib/_co.py)
                bwd.append(call(lib.mpnn_backward_finish_opt, 'backward_finish', slab.data_ptr(), tab.data_ptr(), n_items,
                                t_seg.data_ptr(), self.dsum.data_ptr(), self.dred.data_ptr(), self.S.data_ptr(),
                                self.bn_table.data_ptr(), self.n_bn, t_bno.data_ptr(), self.bn_decay, n, keep_ptr,
                                self.P.data_ptr(), self.A.data_ptr(), self.G.data_ptr(), self.node_stat.data_ptr(),
                                self.hyp.data_ptr(), talr, 1.0 / n, 1.0, self.w_eq.data_ptr() if self.w_eq is not None else None,
                                self.packs.data_ptr(), t_plain.data_ptr(), len(plain)))
                prog['fused_opt'] = True
            else:
                # slab reduction + BatchNorm finalisation (moving averages, dgamma/dbeta): one launch
                bwd.append(call(lib.mpnn_backward_finish, 'backward_finish', slab.data_ptr(), self.G.data_ptr(),
                                tab[6 * first:].data_ptr(), n_items - first, self.dsum.data_ptr(), self.dred.data_ptr(),
                                self.S.data_ptr(), self.bn_table.data_ptr(), self.n_bn, self.bn_decay, n, keep_ptr))
        else:
            bwd.append(call(lib.mpnn_bn_finalize, 'bn_finalize', self.dsum.data_ptr(), self.dred.data_ptr(),
                            self.S.data_ptr(), self.G.data_ptr(), self.bn_table.data_ptr(), self.n_bn,
                            self.bn_decay, n, keep_ptr))
        # member records of the level launches: every pointer is final now
        for mem, cnt, host, rec_bytes, dev_rec in level_fix:
            if self.co_share > 1:
                _hip.check(lib.mpnn_msconv_bwd_level_prepare_rep(mem, cnt, 1, C.cast(host, C.c_void_p)), 'bwd_level records')
            else:
                _hip.check(lib.mpnn_msconv_bwd_level_prepare(mem, cnt, C.cast(host, C.c_void_p)), 'bwd_level records')
            dev_rec.copy_(torch.frombuffer(bytearray(host.raw), dtype=torch.uint8))
        if dp:
            bwd.append(marker('bucket', 'end'))
        return prog


    # ------------------------------------------------------------------ evaluation programs
    def _depths(self):
        depth = {}
        for b in self.blocks:
            depth[id(b)] = 0 if b.parent is None else depth[id(b.parent)] + 1
        return depth


    def _groupable(self):
        return all(c % 16 == 0 and not (c % 64 == 0 and h >= 16) for b in self.blocks for c, h in zip(b.C, b.H))



    def groupable(self):
        """The forward convs of this architecture all have the wavefront-grouped (table-driven) launch form: its nets can share
        launches in a co-trained group (lib/_co.py)."""
        return self._groupable()

    def _route_args(self, n, mode, loss):
        ϕ, kind = self.net.hypers, self.net._net_kind
        ra = _hip.RouteArgs()
        ra.net_type = {'sr': _hip.NET_SR, 'actor': _hip.NET_ACTOR, 'critic': _hip.NET_CRITIC}[kind]
        ra.n_nodes, ra.n_leaves, ra.n_switches, ra.max_sinks = len(self.nodes), len(self.leaves), len(self.switches), self.max_sinks
        ra.optimistic = int(bool(getattr(ϕ, 'optimistic', False)))
        ra.use_cls_err = int(bool(getattr(ϕ, 'use_cls_err', False)))
        ra.want_grad = 1 if mode == 'tr' else 0
        ra.nodes, ra.sw_children, ra.node_ops = self.node_tab.data_ptr(), self.kid_tab.data_ptr(), self.node_ops.data_ptr()
        ra.hyp = self.hyp.data_ptr()
        ra.k_cpt_vec = self.k_cpt.data_ptr() if bool(getattr(ϕ, 'dyn_k_cpt', False)) else None
        ra.r, ra.c_err, ra.d_cor = self.r.data_ptr(), self.c_err.data_ptr(), self.d_cor.data_ptr()
        ra.p_tr, ra.p_ev, ra.w_cerr, ra.dr = self.p_tr.data_ptr(), self.p_ev.data_ptr(), self.w_cerr.data_ptr(), self.dr.data_ptr()
        ra.node_stat = self.node_stat.data_ptr() if mode == 'tr' else None
        if mode == 'tr':
            # more than two workgroups (trees at 128 samples, chains beyond): per-workgroup partial sums + a last-arriver sum in
            # workgroup order instead of fp32 atomics -- the TALR statistics are the same bits from run to run
            need = (n + 15) // 16 * (len(self.nodes) * 2 + 8)          # (+ 4 doubles per workgroup: the loss sums)
            if getattr(self, '_stat_part', None) is None or self._stat_part.numel() < need:
                self._stat_part = torch.zeros(need, device=self.dev)
                self._stat_ticket = torch.zeros(4, dtype=torch.int32, device=self.dev)
            ra.stat_part, ra.stat_ticket = self._stat_part.data_ptr(), self._stat_ticket.data_ptr()
            self._keep += [self._stat_part, self._stat_ticket]
        ra.loss = loss.data_ptr()
        ra.n, ra.n_total = n, n
        self._keep.append(ra)
        return ra
