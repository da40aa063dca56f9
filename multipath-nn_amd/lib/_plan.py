"""Static execution plan of a net over pre-allocated HBM buffers.

``Engine`` turns a linked ``Net`` (lib/net_types.py) into lists of launches of
the HIP kernels behind the C ABI (include/mpnn_hip.h).  Everything a training
step needs lives in a handful of flat device buffers:

* ``P`` / ``A`` / ``G``: parameters, momentum accumulators, gradients (fp32, one
  flat buffer each, same layout; ``G`` carries the per-node TALR statistics at
  its tail so ONE all-reduce serves data-parallel training);
* ``S``: BatchNorm moving averages (non-trainable state);
* per block and scale: the pre-BatchNorm conv sums ``s`` (the only activation
  that is materialised -- BatchNorm+ReLU are applied by consumers on load) and
  one gradient buffer that holds dz, then g, in place;
* fp64 arenas for BatchNorm statistics (forward sums, backward reductions).

There is no CPU fallback: constructing an Engine loads libmpnn_hip.so and needs
a GPU.  PyTorch provides device memory, streams and (optionally) graph capture.

Reference semantics restated here (graph structure only; arithmetic is in the
kernels): tree walk of Net.link (net_types.py:56-63), MultiscaleConvMax's
negative indexing of the input pyramid (layer_types.py:163,181-185), the
parameter -> tree-node map of minimize_expectation (net_types.py:28-34).
"""
import ctypes as C
import os
import unicodedata

import numpy as np
import torch

from lib import _hip
from lib.layer_types import Chain
from lib.net_types import n_leaves, params_list_rec

ROUTER_COMPS = ['Select', 'LinTrans', 'BatchNorm', 'Rect', 'LinTrans', 'BatchNorm', 'Rect', 'LinTrans']
BLOCK_COMPS = ['MultiscaleConvMax', 'MultiscaleBatchNorm', 'MultiscaleRect']
HEAD_COMPS = ['Select', 'LinTrans', 'Softmax', 'CrossEntropyError']
OPT_CHUNK = 2048
# hipGraph capture mode: thread-local, so that other threads' runtime calls (the process group's
# watchdog polling its events under data parallelism) are not errors while this thread captures
CAPTURE_MODE = 'thread_local'


def _nf(name):
    """Python NFKC-normalises identifiers (the keyword ``ϵ=`` U+03F5 is stored as U+03B5) but not
    string literals: every string-keyed attribute lookup must go through the same normalisation."""
    return unicodedata.normalize('NFKC', name)


def _attr(obj, name, default=None):
    return getattr(obj, _nf(name), default)


def _kind(ℓ):
    if isinstance(ℓ, Chain):
        t = [type(c).__name__ for c in ℓ.comps]
        if t == ['ToPyramid']:
            return 'pyramid'
        if t == BLOCK_COMPS:
            return 'block'
        if t == HEAD_COMPS:
            return 'head'
    raise NotImplementedError(
        'layer %r (%s) is outside the MI355X hot path: supported tree nodes are the '
        'ToPyramid, ReConvMax and LogReg chains of arch_and_hypers.py' % (ℓ.name, type(ℓ).__name__))


class _Node:
    pass


class BoundInput:
    """Feed value for ``net.x0`` / ``net.y`` that means "whatever the step's prologue puts into the engine's own
    input buffer" (lib/data.py: Dataset.bind_engine -- the on-device batch assembly is launch 0 of the step).  It
    names the buffer instead of holding a view of it: the buffers are reallocated when a larger batch comes by (the
    statistics pass at 4 096 images), and a view taken before that would feed the step from an orphaned allocation."""

    def __init__(self, eng, which, n):
        self.eng, self.which, self.n = eng, which, int(n)

    @property
    def shape(self):
        return (self.n,) + tuple(getattr(self.eng, self.which).shape[1:])

    def tensor(self):
        return getattr(self.eng, self.which)[:self.n]


class _Block:
    pass


class Engine:
    def __init__(self, net, device=None, n_max=128):
        self.net = net
        self.lib = _hip.load()
        if not torch.cuda.is_available():
            raise _hip.HipError('no GPU visible: the multipath-nn hot path runs on MI355X only')
        if device is None:
            device = 'cuda:%d' % int(os.environ.get('LOCAL_RANK', '0'))
        self.dev = torch.device(device)
        torch.cuda.set_device(self.dev)
        self.n_max = self.n_max_bwd = 0
        self.use_graph = bool(int(os.environ.get('MPNN_GRAPH', '1')))     # hipGraph replay of the step (0: eager launches)
        self.multi_stream = bool(int(os.environ.get('MPNN_STREAMS', '0')))
        self.group_fwd = bool(int(os.environ.get('MPNN_FWD_GROUP', '1')))   # wavefront-grouped forward launches
        self.bwd_levels = bool(int(os.environ.get('MPNN_BWD_LEVELS', '1')))  # one backward launch per dependency level
        self.routed_min_batch = int(os.environ.get('MPNN_ROUTED_MIN_BATCH', '512'))
        self.fold_clear = bool(int(os.environ.get('MPNN_FOLD_CLEAR', '1')))  # no clearing launch in a training step
        self._acc_clean = False          # the step's accumulators (slot sums, TALR statistics, loss) are cleared
        self._streams = []
        self._event_keep = []
        self.n_streams = 1
        self.world = 1
        self.allreduce = None            # callable(G) installed by lib/_dp.py
        self.allreduce_capturable = False  # the collective may be captured into a hipGraph (RCCL on device tensors)
        self.dp_agree = None             # callable(bool) -> bool: logical AND over the ranks (lib/_dp.py)
        self.dp_quiesce = None           # callable(): no collective of the process group is pending or being polled
        self.dp_one_graph = bool(int(os.environ.get('MPNN_DP_ONE_GRAPH', '1')))
        # data parallel with SEVERAL gradient buckets (MPNN_DP_BUCKETS > 1; the default is one, see _alloc_params):
        # compute units the backward launches that run beside a bucket's all-reduce leave to the collective's
        # workgroups (mpnn_set_reserved_cus), and whether each bucket is applied (TALR + momentum) on a side stream as
        # soon as its all-reduce has finished.  Both measured on one GPU with a stand-in for the collective
        # (tools/dp_corunner_probe.py, profiles/r04_dp_corunner.txt) and OFF by default: a co-running kernel does not
        # slow the launches it runs beside (the launches behind the bucket points are the deep small-map ones, which
        # do not fill the chip), the reservation costs 12 us per step, and every additional parallel branch of the step
        # graph stalls the main branch by ~30 us.
        self.dp_reserve_cus = int(os.environ.get('MPNN_DP_RESERVE_CUS', '0'))
        self.dp_bucket_opt = bool(int(os.environ.get('MPNN_DP_BUCKET_OPT', '0')))
        self._opt_stream = None
        self.prologue = None             # callable(stream): first launch of every training step (the input pipeline)
        self.prologue_slot = None        # callable(stream, j): the same for step j of a K-step graph (run_steps)
        # single process: the launch that ends the backward pass also applies the update (mpnn_backward_finish_opt)
        self.fuse_opt = bool(int(os.environ.get('MPNN_FUSE_OPT', '1')))
        # co-training (lib/_co.py): this net shares every launch of its training step with co_share - 1 other nets of the
        # same architecture -- the planner budgets workgroups against resident slots / co_share and every backward
        # launch takes the table-driven (level) form, whose records the co-trainer concatenates over the nets
        self.co_share = 1
        self._keep = []
        self._progs = {}
        self._graphs = {}
        self._classify()
        self._alloc_params()
        self.init_params(net.hypers.__dict__.get('seed'))
        self._ensure_capacity(n_max)
        self.last_n = 0
        self.last_mode = 'ev'

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

    # ------------------------------------------------------------------ running
    def _stage(self, feed, upload_hyp=True):
        net = self.net
        x0 = feed[net.x0]
        n = int(x0.shape[0])
        self._ensure_capacity(n, feed.get(net.mode, net.mode.default) == 'tr')

        def put(dst, src):
            if isinstance(src, BoundInput):
                if src.eng is not self or self.prologue is None:
                    raise ValueError('a BoundInput feeds the engine it was bound to, with the prologue installed')
                return
            if isinstance(src, torch.Tensor):
                if src.data_ptr() == dst.data_ptr():
                    return
                dst.copy_(src.reshape(dst.shape), non_blocking=True)
            else:
                dst.copy_(torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32)).reshape(dst.shape),
                          non_blocking=True)
        put(self.x0[:n], x0)
        put(self.y[:n], feed[net.y])
        h = self._hyp_values(feed, n, put)
        if not upload_hyp:                      # (lib/_co.py uploads the schedule values of all its nets at once)
            self._hyp_sent = None
            return n, feed.get(net.mode, net.mode.default)
        if self._hyp_sent is None or not torch.equal(h, self._hyp_sent):
            # Upload through a ring of pinned buffers: the copy is asynchronous and, under hipGraph
            # replay, the host runs many steps ahead of the stream -- rewriting ONE staging buffer in place
            # would let step t's DMA read the schedule values of step t + k.  A slot is reused only after
            # the event recorded behind its last copy has completed.
            k = self._hyp_slot = (self._hyp_slot + 1) % len(self._hyp_ring)
            buf, ev = self._hyp_ring[k]
            if ev is not None:
                ev.synchronize()
            buf.copy_(h)
            self.hyp.copy_(buf, non_blocking=True)          # (skipped while the schedule holds them constant)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._hyp_ring[k] = (buf, ev)
            self._hyp_sent = h.clone()
            self._hyp_epoch = getattr(self, '_hyp_epoch', 0) + 1       # (lib/_co.py: this row of its buffer was rewritten)
        return n, feed.get(net.mode, net.mode.default)

    def _hyp_values(self, feed, n, put=None):
        """The MPNN_HYP_N schedule / hyper-parameter values of one step (self._hyp_stage, a host tensor); put: stages the
        per-sample k_cpt vector of a dyn_k_cpt net."""
        net = self.net
        ϕ = net.hypers
        get = lambda name, default: feed.get(_attr(net, name), _attr(ϕ, name, default))
        h = self._hyp_stage
        h[_hip.HYP_LR] = float(get('λ_lrn', 0.0))
        h[_hip.HYP_MU] = float(get('μ_lrn', 0.0))
        h[_hip.HYP_TAU] = float(get('τ', 1.0))
        h[_hip.HYP_EPS] = float(get('ϵ', 0.0))
        h[_hip.HYP_KDEC] = float(_attr(ϕ, 'k_dec', 0.0))
        h[_hip.HYP_KCRE] = float(_attr(ϕ, 'k_cre', 0.0))
        h[_hip.HYP_ARTR] = float(_attr(ϕ, 'α_rtr', 1.0))
        if getattr(ϕ, 'dyn_k_cpt', False):
            if put is None:
                raise NotImplementedError('per-sample k_cpt: one step per call')
            k = feed[net.k_cpt]
            k = np.broadcast_to(np.asarray(k, np.float32).reshape(-1), (n,)) if not isinstance(k, torch.Tensor) else k.expand(n)
            put(self.k_cpt[:n], k)
            h[_hip.HYP_KCPT] = 0.0
        else:
            h[_hip.HYP_KCPT] = float(getattr(ϕ, 'k_cpt', 0.0))
        return h

    def _launch(self, ops, sec=0):
        """Run a program section.  Sequential order is a valid topological order; with
        ``multi_stream`` the independent launches of the per-scale dependency DAG go to side
        streams (under graph capture they become parallel branches of the hipGraph)."""
        main = torch.cuda.current_stream()
        if not self.multi_stream:
            try:
                for op in ops:
                    self._set_reserve(getattr(op, 'reserve', 0))
                    op(main.cuda_stream)
            finally:
                self._set_reserve(0)
            return
        # One set of side streams per section: re-forking streams that were already joined inside
        # the same hipGraph capture crashes hipStreamEndCapture (ROCm 7.2).
        while len(self._streams) <= sec:
            self._streams.append(None)
        if self._streams[sec] is None or len(self._streams[sec]) < self.n_streams - 1:
            self._streams[sec] = [torch.cuda.Stream(device=self.dev) for _ in range(self.n_streams - 1)]
        streams = [main] + self._streams[sec][:self.n_streams - 1]
        events, forked = {}, False
        keep = self._event_keep            # events must outlive an open graph capture (HIP)
        for op in ops:
            if op.what == 'fork':
                e = torch.cuda.Event()
                keep.append(e)
                e.record(main)
                for s_ in streams[1:]:
                    s_.wait_event(e)
                forked = True
                continue
            if op.what == 'join':
                # every forked stream rejoins main, used or not (a stream left dangling inside a
                # graph capture is an error)
                for s_ in streams[1:]:
                    e = torch.cuda.Event()
                    keep.append(e)
                    e.record(s_)
                    main.wait_event(e)
                forked = False
                continue
            st = streams[op.stream]
            for w in op.waits:
                ev = events.get(w)
                if ev is not None and ev[0] != op.stream:
                    st.wait_event(ev[1])
            with torch.cuda.stream(st):
                op(st.cuda_stream)
            if op.records:
                e = torch.cuda.Event()
                keep.append(e)
                e.record(st)
                events[op.records] = (op.stream, e)
        assert not forked, 'program section ended with side streams still forked'

    def _set_reserve(self, cus):
        """Compute units the grids of the launches that follow leave free (host-side state of the library)."""
        if cus != getattr(self, '_reserved', 0):
            self.lib.mpnn_set_reserved_cus(cus)
            self._reserved = cus

    def _zero(self, train):
        if train:
            self._zarena.zero_()           # (G lives in the same arena)
        else:
            self._ev_arena.zero_()

    def _pack(self):
        _hip.check(self.lib.mpnn_pack_weights(self.P.data_ptr(), self.packs.data_ptr(), self.pack_desc.data_ptr(),
                                              self.n_pack, torch.cuda.current_stream().cuda_stream), 'pack_weights')

    def _opt(self, n, bucket=None):
        """TALR + L2 + momentum update (net_types.py:24-37) of every parameter, or of one gradient bucket's."""
        talr = 1 if (self.net._net_kind != 'sr' and getattr(self.net.hypers, 'talr', False)) else 0
        first, count = (0, self.n_seg) if bucket is None else self.seg_range[bucket]
        if count == 0:
            return
        _hip.check(self.lib.mpnn_talr_momentum_step(
            self.P.data_ptr(), self.A.data_ptr(), self.G.data_ptr(), self.seg[first * _hip.SEG_INTS:].data_ptr(), count,
            self.node_stat.data_ptr(), self.hyp.data_ptr(), talr, 1.0 / (n * self.world), 1.0 / self.world,
            self.w_eq.data_ptr() if self.w_eq is not None else None, self.packs.data_ptr(),
            torch.cuda.current_stream().cuda_stream), 'talr_momentum_step')

    def _begin(self, train):
        """mpnn_step_begin: pack the weights and clear the step's accumulators in one launch
        (evaluation: loss sums, routed sample counts, r / c_err / d_cor)."""
        z = self._zarena if train else self._ev_arena
        fresh = self._packs_fresh                  # (always true inside a captured graph: run() packs eagerly first)
        _hip.check(self.lib.mpnn_step_begin(self.P.data_ptr(), self.packs.data_ptr(),
                                            None if fresh else self.pack_desc.data_ptr(), 0 if fresh else self.n_pack,
                                            z.data_ptr(), z.numel() * z.element_size(),
                                            torch.cuda.current_stream().cuda_stream), 'step_begin')
        self._packs_fresh = True

    def invalidate_packs(self):
        """The parameters were written from outside a training step (initialisation, Param.assign, a
        checkpoint, a broadcast): the weight packs are rebuilt before the next run.  Inside training the
        optimizer kernel keeps them current.  Call this after writing ``eng.P`` directly."""
        self._packs_fresh = False

    def _sections(self, prog, train):
        """The step as a list of (launches, bucket) sections: a section ends where a gradient bucket
        becomes final (data-parallel programs; bucket = name in self.dp_buckets), the last one has
        bucket None.  Single-process programs are one section."""
        ops = list(prog['fwd']) + (list(prog['bwd']) if train else [])
        out, cur = [], []
        for op in ops:
            if op.what == 'bucket':
                out.append((cur, op.tag))
                cur = []
            else:
                cur.append(op)
        if cur or not out:
            out.append((cur, None))
        return out

    def _reduce_bucket(self, name):
        lo, hi = self.dp_buckets[name]
        return self.allreduce(self.G[lo:hi])

    @staticmethod
    def _wait(handles):
        for h in handles:
            if hasattr(h, 'wait'):
                h.wait()

    def batch_stat_sums(self):
        """fp64 slot sums (sum x, sum x^2 per BatchNorm, layout of the finalize table) of the last training step."""
        return self.dsum_last if getattr(self, '_last_fold', False) else self.dsum

    def _clear_if_needed(self, prog, train):
        """Clear the step's accumulators unless the previous training step left them cleared (programs with
        prog['fold']: slot sums cleared by the launch that ends the backward pass, TALR statistics and loss sums
        by the launch in front of mpnn_route).  Evaluation programs always clear their own arena."""
        if not train:
            self._begin(False)
        elif not (prog.get('fold') and self._acc_clean):
            self._begin(True)

    def _phase_a(self, prog, train, n=None):
        """Everything of a step except the optimizer (eager launches).  Data parallel: the all-reduce
        of each gradient bucket is issued as soon as its section is queued -- lib/_dp.py returns an
        asynchronous handle, so the collective runs on RCCL's stream beside the rest of the backward
        pass; all handles are waited for (a stream-level dependency) before the optimizer."""
        n = prog['n'] if n is None else n
        if train and self.prologue is not None:
            self.prologue(torch.cuda.current_stream().cuda_stream)
        self._clear_if_needed(prog, train)
        if train:
            self._acc_clean = False                # (until the whole backward pass has been queued)
        if not (train and self.allreduce is not None):
            self._launch(prog['fwd'], 0)
            if train:
                self._launch(prog['bwd'], 1)
                self._acc_clean = bool(prog.get('fold'))
            return
        handles = []
        for k, (ops, bucket) in enumerate(self._sections(prog, train)):
            self._launch(ops, k)
            if bucket is not None:
                handles.append((bucket, self._reduce_bucket(bucket)))
                if self._bucket_opt_on():
                    self._opt_bucket(n, *handles.pop())
        self._wait([h for _, h in handles])
        self._acc_clean = bool(prog.get('fold'))

    def _bucket_opt_on(self):
        return self.dp_bucket_opt and len(self.dp_buckets) > 1

    def _opt_bucket(self, n, bucket, handle):
        """Apply one gradient bucket as soon as its all-reduce has finished.  The last bucket: on the compute stream,
        which then also waits for the side stream.  Earlier buckets: on a side stream behind the collective (the
        parameters they update -- exits; the deep blocks' conv weights and their packs -- are not read by the rest of the
        backward pass, and the node statistics every TALR scale needs came with the FIRST bucket), so the compute
        stream goes on with the backward pass and only the `end` bucket's update stays exposed."""
        main = torch.cuda.current_stream()
        last = bucket == list(self.dp_buckets)[-1]
        if last:
            self._wait([handle])
            self._opt(n, bucket)
            if self._opt_stream is not None:
                main.wait_stream(self._opt_stream)
            return
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=self.dev)
        side = self._opt_stream
        with torch.cuda.stream(side):
            if hasattr(handle, 'wait'):
                handle.wait()                      # (a stream dependency on the collective, on the side stream)
            else:
                side.wait_stream(main)             # (a blocking collective: it was ordered on the compute stream)
            self._opt(n, bucket)

    def run(self, feed, train, routed=False):
        if len(self._event_keep) > 4096:
            torch.cuda.synchronize()
            self._event_keep.clear()
        n, mode = self._stage(feed)
        if not self._packs_fresh:                  # (eager, outside any captured graph)
            self._pack()
            self._packs_fresh = True
        if train and mode != 'tr':
            raise ValueError("net.train.run needs net.mode: 'tr' in the feed")
        if not train and mode == 'tr':
            # A fetch in mode 'tr' without the train op: the reference evaluates with BATCH statistics, soft routing
            # p_tr, and moves every consumed BatchNorm's averages as a side effect of the forward pass
            # (layer_types.py:231-236, net_types.py:50-52).  Here: the forward half of the training program, then
            # the moving-average update of the conv BatchNorms (which otherwise rides in the launch that ends the
            # backward pass); the router BatchNorms move theirs in mpnn_exit_tail_fwd.  No gradients, no optimizer.
            self._forward_tr(n)
            self.last_n, self.last_mode = n, mode
            self._bind_views(n)
            return
        prog = self.program(mode, n, routed)
        if train and prog.get('fold') and self._acc_clean and os.environ.get('MPNN_PLAN_DEBUG'):
            # (debug: a step without a clearing launch relies on the previous step having left these cleared)
            torch.cuda.synchronize()
            assert not bool(self.dsum.any()) and not bool(self.dred.any()), 'slot sums not clean at the start of a step'
        if not self.use_graph:
            self._step_eager(prog, train, n)
        else:
            self._run_graphed(prog, train, n)
        self.last_n, self.last_mode = n, mode
        if train:
            self._last_fold = bool(prog.get('fold'))
        self._bind_views(n)

    STEPS_MAX = 8                       # most training steps in one hipGraph (run_steps)

    def run_steps(self, feeds):
        """K training steps as ONE hipGraph replay (K = len(feeds) <= STEPS_MAX; same results as K calls of run()).

        Between two replays of the one-step graph the GPU idles ~8.6 us (profiles/r04_final_step_timeline.txt: host /
        runtime, not kernel time); K steps in one graph pay that once.  What changes from step to step is data, not
        structure: the schedule values (learning rate, temperature) are staged for all K steps at once in a device
        ring and copied into the buffer the step's kernels read by the head workgroup of the step's own
        mpnn_exit_tail_fwd (mpnn_exit_tail_args.hyp_src: no launch of its own); with the input pipeline bound
        (Dataset.bind_engine) launch 0 of step j gathers the batch staged in record slot j.  Without it every feed
        must name the engine's resident input buffers (the same batch K times: the benchmark).  Falls back to K
        single-step calls where the one-graph form does not apply (data parallel, eager).  Per-sample k_cpt vectors
        (dyn_k_cpt nets) ride in a device ring like the schedule values; the launches that read them get per-step records."""
        net, K = self.net, len(feeds)
        dyn = bool(getattr(net.hypers, 'dyn_k_cpt', False))
        # data parallel: only the form in which the whole step -- its collectives included -- is ONE captured graph
        # (lib/_dp.py: RCCL, self-tested); K steps then hold K all-reduces.  The section-graph form issues its collectives
        # from the host between replays and stays one step at a time.
        dp = self.allreduce is not None
        ok = 1 < K <= self.STEPS_MAX and self.use_graph and not self.multi_stream and \
            (not dp or (self.dp_one_graph and self.allreduce_capturable and not self._bucket_opt_on() and getattr(self, '_k_dp_ok', True)))
        if ok:
            xs = [f[net.x0] for f in feeds]
            ys = [f[net.y] for f in feeds]
            # every feed names THIS engine's inputs, x0 and y alike, and one batch size (a mixed list would be captured and
            # replayed with step 0's shapes)
            bound = all(isinstance(x, BoundInput) and isinstance(y, BoundInput) and x.eng is self and y.eng is self
                        for x, y in zip(xs, ys)) and len({x.n for x in xs} | {y.n for y in ys}) == 1
            same = all(isinstance(x, torch.Tensor) and x.data_ptr() == xs[0].data_ptr() and x.shape == xs[0].shape for x in xs) and \
                isinstance(xs[0], torch.Tensor) and xs[0].data_ptr() == self.x0.data_ptr() and \
                all(isinstance(y, torch.Tensor) and y.data_ptr() == self.y.data_ptr() and y.shape[0] == xs[0].shape[0] for y in ys)
            ok = (bound and self.prologue_slot is not None) or (same and self.prologue is None)
            ok = ok and all(f.get(net.mode, net.mode.default) == 'tr' for f in feeds)
        def one_by_one():
            # step by step; with the input pipeline bound, step j must gather from record slot j (the caller staged K slots)
            slots = self.prologue_slot is not None and all(isinstance(f[net.x0], BoundInput) for f in feeds)
            keep_p, keep_g = self.prologue, self.use_graph
            try:
                for j, f in enumerate(feeds):
                    if slots and j > 0:         # (slot 0 is what the one-step graph reads: step 0 takes the usual path)
                        self.prologue, self.use_graph = (lambda st, j=j: self.prologue_slot(st, j)), False
                    self.run(f, True)
            finally:
                self.prologue, self.use_graph = keep_p, keep_g
        if not ok:
            return one_by_one()
        n = int(xs[0].shape[0])
        # (every planner setting that selects the program is part of the key: a graph captured from another program must
        # not be replayed after a switch)
        key = ('trK', n, K, self.bwd_levels, self.fold_clear, self.fuse_opt, self.co_share, dp)
        g = self._graphs.get(key)
        if g is None:
            one_by_one()                                        # (first call: the single-step path loads the code objects)
            self._graphs[key] = 'warm'
            return
        prog = self.program('tr', n)
        if not (prog.get('fold') and (prog.get('fused_opt') or dp)):
            return one_by_one()
        if len(self._event_keep) > 4096:
            torch.cuda.synchronize()
            self._event_keep.clear()
        # the K steps' schedule values: one asynchronous upload through a ring of pinned buffers
        if not hasattr(self, '_hypk'):
            self._hypk = torch.zeros(self.STEPS_MAX, _hip.HYP_N, device=self.dev)
            self._hypk_ring = [(torch.zeros(self.STEPS_MAX, _hip.HYP_N).pin_memory(), None) for _ in range(8)]
            self._hypk_slot = -1
        if dyn:
            # per-sample k_cpt (net_types.py:149-160): step j's vector in slot j of a device ring; the K vectors travel in one
            # upload, and the launches that read them (mpnn_lin_fwd / _bwd: the k_cpt column; mpnn_route) get per-step records
            if getattr(self, '_kck', None) is None or self._kck.shape[1] < self.n_max:
                self._kck = torch.zeros(self.STEPS_MAX, self.n_max, device=self.dev)
                self._kck_ring = [(torch.zeros(self.STEPS_MAX, self.n_max).pin_memory(), None) for _ in range(8)]
                self._kck_slot = -1
                self._graphs = {k: v for k, v in self._graphs.items() if k[0] != 'trK'}
                g = self._graphs.get(key)
                if g is None:
                    one_by_one()
                    self._graphs[key] = 'warm'
                    return
            r = self._kck_slot = (self._kck_slot + 1) % len(self._kck_ring)
            kbuf, kev = self._kck_ring[r]
            if kev is not None:
                kev.synchronize()
            stage, on_dev = [], {}
            for j, f in enumerate(feeds):
                def put_k(dst, src, j=j):
                    if isinstance(src, torch.Tensor) and src.is_cuda:
                        on_dev[j] = src                     # (already on the device: copied there, no host round trip)
                    else:
                        src = src if isinstance(src, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32))
                        kbuf[j, :n].copy_(src.reshape(-1))
                stage.append(self._hyp_values(f, n, put_k).clone())
            hs = torch.stack(stage)
            if not on_dev:
                self._kck[:K, :n].copy_(kbuf[:K, :n], non_blocking=True)
            else:
                for j in range(K):
                    self._kck[j, :n].copy_(on_dev[j].reshape(-1) if j in on_dev else kbuf[j, :n], non_blocking=True)
            kev = torch.cuda.Event()
            kev.record(torch.cuda.current_stream())
            self._kck_ring[r] = (kbuf, kev)
        else:
            hs = torch.stack([self._hyp_values(f, n).clone() for f in feeds])
        if getattr(self, '_hypk_sent', None) is None or self._hypk_sent.shape != hs.shape or not torch.equal(hs, self._hypk_sent):
            r = self._hypk_slot = (self._hypk_slot + 1) % len(self._hypk_ring)
            buf, ev = self._hypk_ring[r]
            if ev is not None:
                ev.synchronize()
            buf[:K].copy_(hs)
            self._hypk[:K].copy_(buf[:K], non_blocking=True)     # (skipped while the K steps' values repeat: constant schedules)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._hypk_ring[r] = (buf, ev)
            self._hypk_sent = hs
        self._hyp_sent = None                                   # (the graph rewrites self.hyp on the device)
        self._hyp_epoch = getattr(self, '_hyp_epoch', 0) + 1
        if not self._packs_fresh:
            self._pack()
            self._packs_fresh = True
        if g == 'warm':
            torch.cuda.synchronize()
            if not self._acc_clean:
                self._begin(True)
                self._acc_clean = True
            ops = [op for op in list(prog['fwd']) + list(prog['bwd']) if op.what not in ('fork', 'join')]
            tails = [op for op in ops if op.what == 'exit_tail_fwd']
            assert len(tails) == 1 and tails[0].host
            tabs = []
            for j in range(K):
                recs = []
                for k, rec in enumerate(tails[0].host):
                    c = type(rec)()
                    C.memmove(C.byref(c), C.byref(rec), C.sizeof(rec))
                    if k == 0:
                        c.hyp_src, c.hyp_dst = self._hypk[j].data_ptr(), self.hyp.data_ptr()
                    recs.append(c)
                tabs.append(_hip.to_device_table(recs, self.dev))
            self._keep += tabs
            ktabs = {}                       # (step, launch) -> the launch's records with step j's k_cpt vector
            if dyn:
                for j in range(K):
                    kp = self._kck[j].data_ptr()
                    for op in ops:
                        if op.what in ('lin_fwd', 'lin_bwd') and getattr(op, 'host', None):
                            recs = []
                            for rec in op.host:
                                c = type(rec)()
                                C.memmove(C.byref(c), C.byref(rec), C.sizeof(rec))
                                if c.k_cpt:
                                    c.k_cpt = kp
                                recs.append(c)
                            ktabs[(j, id(op))] = _hip.to_device_table(recs, self.dev)
                        elif op.what == 'route':
                            c = type(op.host)()
                            C.memmove(C.byref(c), C.byref(op.host), C.sizeof(op.host))
                            c.k_cpt_vec = kp
                            ktabs[(j, id(op))] = c
                self._keep += list(ktabs.values())
            def step_op(j, op):
                """Launch `op` as step j of the graph runs it: its own records where they differ from step to step."""
                if op.what == 'exit_tail_fwd':
                    fn = lambda st: _hip.check(op.fn(tabs[j].data_ptr(), *op.args[1:], st), 'exit_tail_fwd')
                elif (j, id(op)) in ktabs and op.what == 'route':
                    fn = lambda st: _hip.check(op.fn(C.byref(ktabs[(j, id(op))]), st), 'route')
                elif (j, id(op)) in ktabs:
                    fn = lambda st: _hip.check(op.fn(ktabs[(j, id(op))].data_ptr(), *op.args[1:], st), op.what)
                else:
                    return op
                for a in ('what', 'tag', 'flops', 'reserve'):
                    if hasattr(op, a):
                        setattr(fn, a, getattr(op, a))
                return fn
            g = torch.cuda.CUDAGraph()
            if not dp:
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    st = torch.cuda.current_stream().cuda_stream
                    for j in range(K):
                        if self.prologue_slot is not None:
                            self.prologue_slot(st, j)
                        for op in ops:
                            step_op(j, op)(st)
            else:
                # K data-parallel steps, each with its gradient all-reduce(s) on the process group's stream and the optimizer
                # behind them, captured as ONE graph: the one-step form (_run_graphed: `_step_eager` under capture) K times
                # with step j's records.  Every rank must end up with the same form: the ranks agree on the outcome.
                err = None
                if self.dp_quiesce is not None:
                    self.dp_quiesce()
                keep_p = self.prologue
                try:
                    with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                        for j in range(K):
                            if self.prologue_slot is not None:
                                self.prologue = lambda st, j=j: self.prologue_slot(st, j)
                            prog_j = dict(prog, fwd=[step_op(j, op) for op in prog['fwd']], bwd=[step_op(j, op) for op in prog['bwd']])
                            self._step_eager(prog_j, True, n)
                except Exception as e:
                    err = e
                finally:
                    self.prologue = keep_p
                torch.cuda.synchronize()
                agreed = self.dp_agree(err is None) if self.dp_agree is not None else err is None
                if not agreed:
                    import warnings
                    warnings.warn('capturing %d data-parallel steps as one hipGraph failed on some rank (here: %r): one step per '
                                  'replay from now on' % (K, err))
                    self._k_dp_ok = False
                    self._graphs.pop(key, None)
                    self._acc_clean = False
                    return one_by_one()
                self._acc_clean = True
            self._graphs[key] = g
        if not self._acc_clean:                                 # something outside run() left the accumulators dirty
            self._begin(True)
        self._acc_clean = False
        g.replay()
        self._acc_clean = True
        self.last_n, self.last_mode, self._last_fold = n, 'tr', True
        self._bind_views(n)

    def _step_eager(self, prog, train, n):
        """One step as eager launches (also what a whole-step hipGraph captures): everything up to the optimizer, then
        the optimizer -- unless the data-parallel step already applied every bucket behind its all-reduce."""
        self._phase_a(prog, train, n)
        if train and not prog.get('fused_opt') and not (self.allreduce is not None and self._bucket_opt_on()):
            self._opt(n)

    def set_prologue(self, fn, fn_slot=None):
        """fn(stream) becomes the first launch of every training step -- lib/data.py installs the on-device batch
        assembly (mpnn_augment_batch) here, so that it is replayed with the step's hipGraph.  fn_slot(stream, j): the
        same for step j of a K-step graph (run_steps), reading the records staged in slot j."""
        self.prologue, self.prologue_slot = fn, fn_slot
        self._graphs = {k: v for k, v in self._graphs.items() if k[0] not in ('tr', 'trK')}

    def mark_dirty(self):
        """Tell the engine that something outside run() launched program ops or wrote the step's accumulators (slot
        sums, TALR statistics, loss sums): the next training step starts with a clearing launch instead of relying on
        the previous step having left them cleared."""
        self._acc_clean = False

    def _forward_tr(self, n):
        prog = self.program('tr', n)
        fold = bool(prog.get('fold'))
        if not (fold and self._acc_clean):
            self._begin(True)
        self._acc_clean = False
        self._launch([op for op in prog['fwd'] if op.what not in ('fork', 'join')] if not self.multi_stream else prog['fwd'], 0)
        _hip.check(self.lib.mpnn_bn_finalize(self.dsum.data_ptr(), None, self.S.data_ptr(), None, self.bn_table.data_ptr(),
                                             self.n_bn, self.bn_decay, n, self.dsum_last.data_ptr() if fold else None,
                                             torch.cuda.current_stream().cuda_stream), 'bn_finalize')
        self._last_fold = fold
        self._acc_clean = fold

    def _run_graphed(self, prog, train, n):
        """First call runs eagerly (loads code objects); the second captures the step as ONE hipGraph (one process, or
        data parallel over a backend whose collectives capture) or as one graph per gradient-bucket section with the
        collectives issued from the host in between; later calls replay."""
        key = (prog['mode'], n, train, prog.get('routed', False), self.bwd_levels, self.fold_clear,
               self.allreduce is not None, self._bucket_opt_on())
        g = self._graphs.get(key)
        if g is None:
            self._step_eager(prog, train, n)
            self._graphs[key] = 'warm'
            return
        dp = train and self.allreduce is not None
        fold = train and bool(prog.get('fold'))
        if g == 'warm':
            torch.cuda.synchronize()
            if fold and not self._acc_clean:           # (captured without a clearing launch: start from cleared accumulators)
                self._begin(True)
                self._acc_clean = True
            g = None
            if not dp or (self.dp_one_graph and self.allreduce_capturable):
                # One process: ONE graph per step.  Data parallel over RCCL: the WHOLE step -- sections, the asynchronous
                # bucket all-reduces on the process group's stream, the per-bucket updates behind them, the waits --
                # is ONE hipGraph as well (RCCL collectives capture; the collective stream and the update stream become
                # parallel branches of the graph): one replay per step instead of four graph launches and three
                # collective calls from the host.
                err = None
                if dp and self.dp_quiesce is not None:
                    self.dp_quiesce()                      # (the watchdog must not poll an eager collective during the capture)
                try:
                    ga = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(ga, capture_error_mode=CAPTURE_MODE):
                        self._step_eager(prog, train, n)
                    g = ([(ga, None)], 'whole')
                except Exception as e:
                    if not dp:
                        raise
                    err = e
                if dp:
                    # EVERY rank must replay the same form (a rank that fell back issues its collectives from the
                    # host, the others inside their graphs): the ranks agree on the outcome of the capture
                    torch.cuda.synchronize()
                    ok = self.dp_agree(err is None) if self.dp_agree is not None else err is None
                    if not ok:
                        import warnings
                        warnings.warn('capturing the data-parallel step as one hipGraph failed on some rank (here: %r): '
                                      'every rank falls back to one graph per gradient-bucket section' % (err,))
                        g = None
                        self.dp_one_graph = False
                        self._acc_clean = False
                        if fold:
                            self._begin(True)
                            self._acc_clean = True
            if g is None:
                # data parallel: one graph per section (the step up to the point where a gradient bucket
                # is final), the bucket's all-reduce issued between the replays, and a graph for the optimizer
                secs = []
                for k, (ops, bucket) in enumerate(self._sections(prog, train)):
                    gk = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gk, capture_error_mode=CAPTURE_MODE):
                        if k == 0:
                            if train and self.prologue is not None:    # (the input pipeline: launch 0 of the step in every form)
                                self.prologue(torch.cuda.current_stream().cuda_stream)
                            self._clear_if_needed(prog, train)
                        self._launch(ops, k)
                    secs.append((gk, bucket))
                gb = None
                if not self._bucket_opt_on():
                    gb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gb, capture_error_mode=CAPTURE_MODE):
                        self._opt(n)
                g = (secs, gb)
            self._graphs[key] = g
        secs, gb = g
        if fold and not self._acc_clean:               # something outside run() left the accumulators dirty
            self._begin(True)
        if train:
            self._acc_clean = False
        handles = []
        for gk, bucket in secs:
            gk.replay()
            if bucket is not None:
                h = self._reduce_bucket(bucket)
                if self._bucket_opt_on():
                    self._opt_bucket(n, bucket, h)         # (eager launches behind the collective)
                else:
                    handles.append(h)
        if dp and gb != 'whole' and gb is not None:
            self._wait(handles)
            gb.replay()
        if train:
            self._acc_clean = fold

    def time_step_ops(self, mode, n, reps=10):
        """In-situ per-launch timing: whole steps run eagerly (no graph), every launch bracketed by
        HIP events on the launch stream, so each kernel sees the cache state and predecessors it has in
        a real step.  Returns [(what, tag, flops, mean_ms)] in launch order."""
        prog = self.program(mode, n)
        train = mode == 'tr'
        ops = [o for o in list(prog['fwd']) + (list(prog['bwd']) if train else []) if o.what not in ('fork', 'join', 'bucket')]
        st = torch.cuda.current_stream()
        tot = [0.0] * len(ops)
        for rep in range(reps + 1):
            self._begin(train)
            evs = []
            for op in ops:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); op(st.cuda_stream); e1.record(st)
                evs.append((e0, e1))
            torch.cuda.synchronize()
            if rep:                               # first pass warms code objects
                for k, (e0, e1) in enumerate(evs):
                    tot[k] += e0.elapsed_time(e1)
        self._acc_clean = False
        return [(op.what, op.tag, op.flops, t / reps) for op, t in zip(ops, tot)]

    def time_family_blocks(self, mode, n, reps=10):
        """In-situ timing of each maximal run of consecutive launches of one kind (e.g. the 20
        bwd_scale launches of a step) with ONE HIP-event pair around the run: the launches queue
        back to back on the stream, so run time / launches is the mean kernel duration as a profiler
        sees it (per-launch event pairs add the host's launch latency to every kernel).
        Returns {what: (launches, flops, mean_ms_per_step)}."""
        prog = self.program(mode, n)
        train = mode == 'tr'
        ops = [o for o in list(prog['fwd']) + (list(prog['bwd']) if train else []) if o.what not in ('fork', 'join', 'bucket')]
        runs = []
        for op in ops:
            if runs and runs[-1][0] == op.what:
                runs[-1][1].append(op)
            else:
                runs.append((op.what, [op]))
        st = torch.cuda.current_stream()
        acc = {}
        for rep in range(reps + 1):
            self._begin(train)
            evs = []
            for what, group in runs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for op in group:
                    op(st.cuda_stream)
                e1.record(st)
                evs.append((what, group, e0, e1))
            torch.cuda.synchronize()
            if rep:
                for what, group, e0, e1 in evs:
                    a = acc.setdefault(what, [0, 0.0, 0.0])
                    a[0] += len(group); a[1] += sum(o.flops for o in group); a[2] += e0.elapsed_time(e1)
        self._acc_clean = False
        return {k: (v[0] // reps, v[1] / reps, v[2] / reps) for k, v in acc.items()}

    def time_ops(self, mode, n, reps=20):
        """Per-launch timing with HIP events on the launch stream (torch's current
        stream is the stream every kernel of the plan is launched on).  Returns
        [(what, tag, flops, mean_ms)] for one (mode, n) program, forward then backward."""
        prog = self.program(mode, n)
        ops = [o for o in list(prog['fwd']) + (list(prog['bwd']) if mode == 'tr' else []) if o.what not in ('fork', 'join')]
        st = torch.cuda.current_stream()
        out = []
        self._acc_clean = False
        self._zero(mode == 'tr')
        self._pack()
        for op in ops:                        # state made valid by running the whole step once
            op(st.cuda_stream)
        for op in ops:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            op(st.cuda_stream)
            e0.record(st)
            for _ in range(reps):
                op(st.cuda_stream)
            e1.record(st)
            e1.synchronize()
            out.append((op.what, op.tag, op.flops, e0.elapsed_time(e1) / reps))
        return out

    # ------------------------------------------------------------------ results
    def _bind_views(self, n):
        # (the views name persistent buffers: they stay valid until the batch size or the buffers change -- rebuilding them
        # after every step cost the host 80 us per net, a third of a co-trained group's GPU time per step)
        key = (n, getattr(self, '_gen', 0))
        if getattr(self, '_views_key', None) == key:
            return
        self._views_key = key
        nn, nl, MS = len(self.nodes), len(self.leaves), self.max_sinks
        ptr, pev = self.p_tr[:nn * n].view(nn, n), self.p_ev[:nn * n].view(nn, n)
        cerr, dcor = self.c_err[:nl * n].view(nl, n), self.d_cor[:nl * n].view(nl, n)
        for nd in self.nodes:
            ℓ = nd.layer
            ℓ.p_tr, ℓ.p_ev = ptr[nd.idx], pev[nd.idx]
            if hasattr(nd, 'leaf_id') and nd.kind == 'head':
                ℓ.c_err, ℓ.δ_cor = cerr[nd.leaf_id], dcor[nd.leaf_id]
            if hasattr(nd, 'switch_id'):
                sw = nd.switch_id
                ℓ.router.x = self.r[sw * n * MS:(sw + 1) * n * MS].view(n, MS)[:, :len(ℓ.sinks)]

    def state(self):
        """Per-sample statistics of the last run (scripts/train-nets:117-130)."""
        net, n = self.net, self.last_n
        y = self.y[:n]
        out = {}
        leaves = [nd.layer for nd in self.leaves]
        out[(net, 'acc')] = sum(ℓ.p_ev * ℓ.δ_cor for ℓ in leaves)
        out[(net, 'moc')] = sum(nd.layer.p_ev * self.node_ops_host[nd.idx] for nd in self.nodes)
        for ℓ in leaves:
            out[(ℓ, 'p_cor')] = ℓ.p_ev * ℓ.δ_cor
            out[(ℓ, 'p_inc')] = ℓ.p_ev * (1 - ℓ.δ_cor)
            out[(ℓ, 'p_cor_by_cls')] = (ℓ.p_ev * ℓ.δ_cor)[:, None] * y
            out[(ℓ, 'p_inc_by_cls')] = (ℓ.p_ev * (1 - ℓ.δ_cor))[:, None] * y
            if net._net_kind != 'sr':
                out[(ℓ, 'p_tr')] = ℓ.p_tr
            out[(ℓ, 'c_err')] = ℓ.c_err
        for nd in self.switches:
            out[(nd.layer, 'x_rte')] = nd.layer.router.x.abs().mean(1)
        return out
