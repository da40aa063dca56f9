"""Co-training: the nets of one experiment advance TOGETHER, one launch per layer for all of them.

The reference trains the nets of an experiment one after another, each at batch 128 (scripts/train-nets:81-88,159-164:
eight `ac_chain(k_cpt=k)` of one architecture).  One net x 128 images cannot fill 256 compute units: three quarters of
every launch of its step are ramp (dispatch, coefficient tables, first tile), which is why the single-net step sits at
0.20 of the fp32 MFMA peak while the same kernels reach 0.50 at 4 096 images.  The batch-wide BatchNorm
statistics put a grid-wide dependency between consecutive layers, so the depth of a step cannot shrink -- but its WIDTH
can: K nets of one architecture have the same launch list, and launch j of every net is independent of launch j of the
others.

``CoTrainer(nets)`` keeps one ``Engine`` per net -- its own parameters, momentum, gradients, BatchNorm statistics,
activations and batch of 128: reference semantics per net, untouched -- and builds ONE program whose launch j is launch j
of all K nets: the table-driven kernels take the concatenated records (`mpnn_msconv_fwd_group_rep`,
`mpnn_msconv_bwd_level_rep`, `mpnn_route_multi`, `mpnn_backward_finish_opt_multi`; the exit-path kernels already take any
number of records).  The dependency depth stays 33 launches; the work per launch grows K-fold.  The program replays as
one hipGraph.

What changes per net against training it alone: the planner budgets every net's workgroups against resident slots / K,
so the pixel split of the weight gradients (the slabs) differs and the conv gradients agree with the solo step to fp32
summation order, not bit for bit (tests/test_cotrain.py holds them to the whole-net tolerances and to the oracle's).
"""
import ctypes as C

import torch

from lib import _hip
from lib._plan import CAPTURE_MODE


class CoTrainer:
    def __init__(self, nets, share=None):
        if len(nets) < 1:
            raise ValueError('CoTrainer needs at least one net')
        self.nets = list(nets)
        self.engs = [net.engine() for net in self.nets]
        e0 = self.engs[0]
        for e in self.engs:
            if type(e).__name__ != 'Engine':
                raise NotImplementedError('co-training covers the multiscale chain / tree engine (lib/_plan.py)')
            if e.allreduce is not None or e.multi_stream or e.generic_exits or not e.fuse_opt or not e.bwd_levels or not e.group_fwd:
                raise NotImplementedError('co-training runs the single-process, single-stream, fused-optimizer schedule '
                                          'with the tuned exit kernels')
            if e.dev != e0.dev:
                raise ValueError('co-trained nets live on one device')
        self.lib, self.dev = e0.lib, e0.dev
        self.K = len(self.nets)
        # the planner budgets every net's grids against resident slots / share: K when the group has the GPU to itself;
        # larger when several groups run side by side on streams of their own (train-nets --co-train K --streams S)
        self.share = self.K if share is None else int(share)
        if self.K > 1 and self.share < 2:
            raise ValueError('a group of several nets runs the table-driven launch forms: share >= 2')
        # the nets' schedule / hyper-parameter values side by side in ONE device buffer: one upload per joint step instead of
        # one per net (the learning rate changes every step).  The engines' own `hyp` become rows of it; programs that
        # baked the old pointers in are dropped.
        self.hyp_all = torch.zeros(self.K, _hip.HYP_N, device=self.dev)
        for k, e in enumerate(self.engs):
            self.hyp_all[k].copy_(e.hyp)
            e.hyp = self.hyp_all[k]
            e.hyp_rewritten()
            e.drop_programs()
        self._hyp_ring = [(torch.zeros(self.K, _hip.HYP_N).pin_memory(), None) for _ in range(8)]
        self._hyp_slot, self._hyp_sent, self._hyp_epochs = -1, None, None
        self.use_graph = e0.use_graph
        self._progs, self._graphs, self._keep, self._gens = {}, {}, [], None
        self.prologue = None             # callable(stream): ONE launch that assembles every net's batch (Dataset.bind_cotrainer)
        self.prologue_slot = None        # callable(stream, j): the same for step j of a K-step joint graph

    # ------------------------------------------------------------------ the merged program
    def _program(self, n):
        gens = tuple(e.generation for e in self.engs)
        if self._progs and self._gens != gens:             # an engine reallocated its buffers (a larger batch came by)
            self.invalidate()
        if n in self._progs:
            return self._progs[n]
        K, lib, keep = self.K, self.lib, self._keep
        progs = []
        for e in self.engs:
            e.co_share = self.share                        # (a planner setting of THIS program only: the net's solo programs keep 1)
            try:
                progs.append(e.program('tr', n))
            finally:
                e.co_share = 1
        self._gens = tuple(e.generation for e in self.engs)
        skip = ('fork', 'join')
        lists = [[op for op in list(p['fwd']) + list(p['bwd']) if op.what not in skip] for p in progs]
        sig = [[(op.what, op.tag) for op in ops] for ops in lists]
        if any(s != sig[0] for s in sig[1:]) or any(bool(p.get('fold')) != bool(progs[0].get('fold')) for p in progs) or \
                not all(p.get('fused_opt') for p in progs):
            raise NotImplementedError('co-trained nets must have the same architecture (identical launch lists)')

        def launch_of(fn, what, flops, tag, *args):
            def launch(st):
                _hip.check(fn(*args, st), what)
            launch.what, launch.flops, launch.tag, launch.args = what, flops, tag, args
            return launch

        def table(records):
            t = _hip.to_device_table(records, self.dev)
            keep.append(t)
            return t

        merged = []
        for j, ops in enumerate(zip(*lists)):
            o0 = ops[0]
            what, tag, flops = o0.what, o0.tag, sum(o.flops for o in ops)
            if K == 1:
                merged.append(o0)                          # (a group of one: the net's own launches)
            elif what == 'fwd_group':
                cnt = o0.args[2]
                arr = (_hip.ConvFwdArgs * (cnt * K))()
                for r, o in enumerate(ops):
                    for k in range(cnt):
                        arr[r * cnt + k] = o.args[0][k]
                dev = table(list(arr))
                keep.append(arr)
                merged.append(launch_of(lib.mpnn_msconv_fwd_group_rep, what, flops, tag, arr, dev.data_ptr(), cnt, K, self.share))
            elif what in ('lin_fwd', 'exit_tail_fwd', 'exit_tail_bwd', 'lin_bwd'):
                recs = [r for o in ops for r in o.host]
                dev = table(recs)
                merged.append(launch_of(o0.fn, what, flops, tag, dev.data_ptr(), len(recs), *o0.args[2:]))
                merged[-1].fn, merged[-1].host = o0.fn, recs             # (run_steps builds per-step copies of this table)
                merged[-1].first = [sum(len(q.host) for q in ops[:r]) for r in range(K)]      # net r's first record
            elif what == 'route':
                arr = (_hip.RouteArgs * K)(*[o.host for o in ops])
                dev = table(list(arr))
                keep.append(arr)
                merged.append(launch_of(lib.mpnn_route_multi, what, flops, tag, arr, dev.data_ptr(), K))
            elif what == 'bwd_scale':
                if o0.fn is not lib.mpnn_msconv_bwd_level_rep:
                    raise NotImplementedError('co-training: a backward launch is not in its table-driven form')
                cnt = o0.args[1]
                mem = (_hip.BwdMember * (cnt * K))()
                for r, o in enumerate(ops):
                    for k in range(cnt):
                        mem[r * cnt + k] = o.args[0][k]
                rec_bytes = lib.mpnn_msconv_bwd_level_record_size()
                host = (C.c_char * (rec_bytes * cnt * K))()
                _hip.check(lib.mpnn_msconv_bwd_level_prepare_rep(mem, cnt, K, C.cast(host, C.c_void_p)), 'bwd_level records (co-training)')
                dev = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.dev)
                keep += [mem, dev]
                merged.append(launch_of(lib.mpnn_msconv_bwd_level_rep, what, flops, tag, mem, cnt, K, dev.data_ptr()))
            elif what == 'backward_finish':
                if len({float(e.bn_decay) for e in self.engs}) > 1:
                    raise NotImplementedError('co-training: the nets of a group must share the conv BatchNorms\' decay')
                arr = (_hip.FinishNet * K)(*[p['finish_net'] for p in progs])
                dev = table(list(arr))
                keep.append(arr)
                merged.append(launch_of(lib.mpnn_backward_finish_opt_multi, what, flops, tag, arr, dev.data_ptr(), K,
                                        float(self.engs[0].bn_decay)))
            else:
                raise NotImplementedError('co-training: launch %r has no multi-net form' % what)
        prog = dict(ops=merged, n=n, fold=bool(progs[0].get('fold')))
        self._progs[n] = prog
        return prog

    # ------------------------------------------------------------------ running
    def _eager(self, prog):
        st = torch.cuda.current_stream().cuda_stream
        if self.prologue is not None:
            self.prologue(st)
        for e in self.engs:
            if self.prologue is None and e.prologue is not None:
                e.prologue(st)
            e.begin_step(prog['fold'])
        for op in prog['ops']:
            op(st)
        for e in self.engs:
            e.end_step(prog['n'], prog['fold'])

    def run(self, feeds):
        """One training step of every net: feeds[i] is net i's feed (as for ``net.train.run``)."""
        if len(feeds) != self.K:
            raise ValueError('one feed per co-trained net')
        ns = set()
        hs = torch.empty(self.K, _hip.HYP_N)
        for k, (e, net, feed) in enumerate(zip(self.engs, self.nets, feeds)):
            n, mode, row = e.stage_feed(feed)
            hs[k].copy_(row)
            if mode != 'tr':
                raise ValueError("co-training needs net.mode: 'tr' in every feed")
            ns.add(n)
            e.fresh_packs()
        epochs = tuple(e.hyp_epoch for e in self.engs)      # (a net that stepped alone rewrote its row)
        if self._hyp_sent is None or epochs != self._hyp_epochs or not torch.equal(hs, self._hyp_sent):
            self._hyp_epochs = epochs
            r = self._hyp_slot = (self._hyp_slot + 1) % len(self._hyp_ring)
            buf, ev = self._hyp_ring[r]
            if ev is not None:
                ev.synchronize()
            buf.copy_(hs)
            self.hyp_all.copy_(buf, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._hyp_ring[r] = (buf, ev)
            self._hyp_sent = hs
        if len(ns) != 1:
            raise ValueError('co-trained nets step on batches of one size')
        n = ns.pop()
        prog = self._program(n)
        g = self._graphs.get(n) if self.use_graph else None
        if not self.use_graph or g is None:
            self._eager(prog)                              # (first call: loads the code objects)
            if self.use_graph:
                self._graphs[n] = 'warm'
        else:
            if g == 'warm':
                torch.cuda.synchronize()
                for e in self.engs:                        # (captured without clearing launches)
                    if prog['fold']:
                        e.clear_for_capture()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                    self._eager(prog)
                self._graphs[n] = g
            for e in self.engs:
                if prog['fold']:
                    e.begin_step(True)                     # (a clearing launch only if something outside run() left them dirty)
                else:
                    e.mark_dirty()                         # (the graph holds the program's own clearing launch)
            g.replay()
        for e in self.engs:
            e.end_step(n, prog['fold'])

    def set_prologue(self, fn, fn_slot=None):
        """fn(stream): the first launch of every joint step (replaces the nets' own prologues in the joint graph);
        fn_slot(stream, j): the same for step j of a K-step joint graph (run_steps), reading record slot j."""
        self.prologue, self.prologue_slot = fn, fn_slot
        self._graphs.clear()

    STEPS_MAX = 8                       # most joint steps in one hipGraph (as Engine.STEPS_MAX: the record slots of lib/data.py)

    def run_steps(self, feeds_k):
        """S joint training steps as ONE hipGraph replay: feeds_k[j][i] is net i's feed at step j (same results, bit for
        bit, as S calls of run()).  What changes from step to step is data: the S x K rows of schedule values travel in
        one upload into a device ring and are copied into every net's `hyp` row by the head workgroup of the net's first
        record of the step's own mpnn_exit_tail_fwd launch (mpnn_exit_tail_args.hyp_src / hyp_dst, as Engine.run_steps);
        with the input pipeline bound (Dataset.bind_cotrainer) launch 0 of step j gathers every net's batch from record
        slot j (Dataset.stage_cotrainer_draws_k).  Falls back to S calls of run() where the form does not apply (eager
        launches, per-sample k_cpt, feeds that are neither bound nor the engines' resident buffers)."""
        from lib._plan import BoundInput
        S, K = len(feeds_k), self.K
        if any(len(f) != K for f in feeds_k):
            raise ValueError('one feed per co-trained net and step')

        def one_by_one():
            slots = getattr(self, 'prologue_slot', None) is not None and \
                all(isinstance(f[net.x0], BoundInput) for fs in feeds_k for net, f in zip(self.nets, fs))
            keep_p, keep_g = self.prologue, self.use_graph
            try:
                for j, fs in enumerate(feeds_k):
                    if slots and j > 0:             # (slot 0 is what the one-step graph reads)
                        self.prologue, self.use_graph = (lambda st, j=j: self.prologue_slot(st, j)), False
                    self.run(fs)
            finally:
                self.prologue, self.use_graph = keep_p, keep_g
        ok = 1 < S <= self.STEPS_MAX and self.use_graph
        ok = ok and not any(bool(getattr(net.hypers, 'dyn_k_cpt', False)) for net in self.nets)
        if ok:
            for fs in feeds_k:
                for net, e, f in zip(self.nets, self.engs, fs):
                    x, y = f[net.x0], f[net.y]
                    bound = isinstance(x, BoundInput) and isinstance(y, BoundInput) and x.eng is e and y.eng is e and \
                        getattr(self, 'prologue_slot', None) is not None
                    same = isinstance(x, torch.Tensor) and isinstance(y, torch.Tensor) and x.data_ptr() == e.x0.data_ptr() and \
                        y.data_ptr() == e.y.data_ptr() and self.prologue is None and e.prologue is None
                    ok = ok and (bound or same) and f.get(net.mode, net.mode.default) == 'tr'
            ns = {int(f[net.x0].shape[0]) for fs in feeds_k for net, f in zip(self.nets, fs)}
            ok = ok and len(ns) == 1
        if not ok:
            return one_by_one()
        n = ns.pop()
        key = ('K', n, S)
        # (the merged program FIRST: it drops every joint graph when an engine has reallocated its buffers since -- a larger
        # evaluation batch came by -- and a graph looked up before that would be replayed over the old buffers)
        prog = self._program(n)
        g = self._graphs.get(key)
        if g is None:
            one_by_one()                                        # (first call: loads the code objects, settles the capacities)
            self._graphs[key] = 'warm'
            return
        if not prog['fold']:
            return one_by_one()
        # the S x K rows of schedule values: one asynchronous upload through a ring of pinned buffers
        if not hasattr(self, '_hypk'):
            self._hypk = torch.zeros(self.STEPS_MAX, K, _hip.HYP_N, device=self.dev)
            self._hypk_ring = [(torch.zeros(self.STEPS_MAX, K, _hip.HYP_N).pin_memory(), None) for _ in range(8)]
            self._hypk_slot, self._hypk_sent = -1, None
        hs = torch.stack([torch.stack([e.schedule_values(f, n) for e, f in zip(self.engs, fs)]) for fs in feeds_k])
        if self._hypk_sent is None or self._hypk_sent.shape != hs.shape or not torch.equal(hs, self._hypk_sent):
            r = self._hypk_slot = (self._hypk_slot + 1) % len(self._hypk_ring)
            buf, ev = self._hypk_ring[r]
            if ev is not None:
                ev.synchronize()
            buf[:S].copy_(hs)
            self._hypk[:S].copy_(buf[:S], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._hypk_ring[r] = (buf, ev)
            self._hypk_sent = hs
        self._hyp_sent = None                                   # (the graph rewrites every net's hyp row on the device)
        for e in self.engs:
            e.hyp_rewritten()
            e.fresh_packs()
        if g == 'warm':
            torch.cuda.synchronize()
            for e in self.engs:                                 # (captured without clearing launches)
                e.clear_for_capture()
            tails = [op for op in prog['ops'] if op.what == 'exit_tail_fwd']
            assert len(tails) == 1 and getattr(tails[0], 'host', None)
            tail = tails[0]
            first = getattr(tail, 'first', [0])
            tabs = []
            for j in range(S):
                recs = []
                for i, rec in enumerate(tail.host):
                    c = type(rec)()
                    C.memmove(C.byref(c), C.byref(rec), C.sizeof(rec))
                    if i in first:
                        k = first.index(i)
                        c.hyp_src, c.hyp_dst = self._hypk[j, k].data_ptr(), self.engs[k].hyp.data_ptr()
                    recs.append(c)
                tabs.append(_hip.to_device_table(recs, self.dev))
            self._keep += tabs
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                st = torch.cuda.current_stream().cuda_stream
                for j in range(S):
                    if getattr(self, 'prologue_slot', None) is not None:
                        self.prologue_slot(st, j)
                    for op in prog['ops']:
                        if op is tail:
                            _hip.check(op.fn(tabs[j].data_ptr(), *op.args[1:], st), 'exit_tail_fwd')
                        else:
                            op(st)
            self._graphs[key] = g
        for e in self.engs:
            e.begin_step(True)                                  # (a clearing launch only if something outside left them dirty)
        g.replay()
        for e in self.engs:
            e.end_step(n, True)

    def invalidate(self):
        """Drop the merged programs and graphs (an engine reallocated its buffers: a larger batch came by)."""
        self._progs.clear()
        self._graphs.clear()


class CoGroups:
    """Several co-trained groups SIDE BY SIDE: each group's joint hipGraph replays on a stream of its own.

    A joint step of one group is still a chain of 33 dependent launches, each with its ramp and its drain; a second group
    on another hardware queue fills them.  There are no edges between the graphs (the 20-30 us a cross-stream graph edge
    costs, DESIGN.md §6, is never paid), and every group budgets its grids against resident slots / share with the
    groups together asking for about TWICE the resident slots -- the dispatcher keeps the compute units busy from
    whichever queue has a workgroup ready (tools/streams_probe.py: 8 nets as 4 groups of 2, share 4: 1.07 x one group of
    8; 16 nets as 4 groups of 4, share 8: 1.10 x one group of 16; more streams than the 4 hardware queues: slower).

    Groups of ONE net run the net's own launch list (any architecture): the statically-routed chains of the *-sr
    experiments, which differ in depth and cannot share launches, still run side by side.

    ``run(feeds)`` does not wait: the groups free-run against each other from step to step.  ``join()`` makes the
    caller's stream wait for all of them (before a statistics pass, a checkpoint, any read of the nets' state).
    """
    def __init__(self, nets, sizes, streams=4, share=None):
        # streams: how many, or a list of torch streams (plan() measures which ones run side by side before it splits)
        given = None if isinstance(streams, int) else list(streams)
        streams = len(given) if given is not None else int(streams)
        if sum(sizes) != len(nets) or min(sizes) < 1:
            raise ValueError('group sizes must add up to the number of nets')
        self.nets = list(nets)
        K = len(self.nets)
        if share is None:
            # groups that run at once ask together for about twice the resident slots (measured: the sweet spot for 4, 8 and
            # 16 nets, groups of 1, 2 and 4); a lone group has the GPU to itself
            share = -(-max(sizes) * min(len(sizes), int(streams)) // 2) if len(sizes) > 1 else K
        self.share = max(int(share), 2 if max(sizes) > 1 else 1)
        self.groups, self.spans, at = [], [], 0
        for s in sizes:
            self.groups.append(CoTrainer(self.nets[at:at + s], share=self.share))
            self.spans.append((at, at + s))
            at += s
        self.dev = self.groups[0].dev
        self.n_streams = max(1, min(int(streams), len(self.groups)))
        if len(self.groups) == 1:
            self.streams = [None]
        else:
            self.streams = (given if given is not None else concurrent_streams(self.dev, self.n_streams))[:self.n_streams]
        self._forked = False

    @classmethod
    def plan(cls, nets, streams=4, share=None):
        """Split nets (in order) into runs of one architecture, and those into about `streams` equal groups -- `streams` capped
        by the number of streams that really run side by side on this device (concurrent_streams: with fewer hardware
        queues than groups the groups would serialise, and one joint graph is the better form)."""
        found = None
        if streams > 1 and len(nets) > 1:
            found = concurrent_streams(nets[0].engine().dev, streams)
            streams = len(found)
        sig = _arch_signature
        runs, sizes = [], []
        for net in nets:
            s = sig(net)
            if runs and runs[-1][0] == s:
                runs[-1][1] += 1
            else:
                runs.append([s, 1, net])
        K = len(nets)
        for _, cnt, first in runs:
            eng = first.engine()
            if cnt > 1 and hasattr(eng, 'groupable') and not eng.groupable():
                # (an architecture whose forward convs are single launches -- 64+ channels on 16x16 / 32x32 maps -- has no
                # multi-net launch form: its nets run side by side as groups of one)
                sizes += [1] * cnt
                continue
            g = max(1, min(cnt, round(streams * cnt / K)))
            base, extra = divmod(cnt, g)
            sizes += [base + (1 if i < extra else 0) for i in range(g)]
        return cls(nets, sizes, found if (found and len(sizes) > 1) else streams, share)

    def stream_of(self, g):
        return self.streams[g % len(self.streams)]

    def _fork(self):
        if not self._forked and self.streams[0] is not None:
            main = torch.cuda.current_stream()
            for s in self.streams:
                s.wait_stream(main)
        self._forked = True

    def on_group_streams(self, fn):
        """fn(g, group, (lo, hi)) for every group, on the group's stream."""
        self._fork()
        for g, (co, span) in enumerate(zip(self.groups, self.spans)):
            s = self.stream_of(g)
            if s is None:
                fn(g, co, span)
            else:
                with torch.cuda.stream(s):
                    fn(g, co, span)

    def run(self, feeds):
        if len(feeds) != len(self.nets):
            raise ValueError('one feed per net')
        self.on_group_streams(lambda g, co, span: co.run(feeds[span[0]:span[1]]))

    def join(self):
        if self._forked and self.streams[0] is not None:
            main = torch.cuda.current_stream()
            for s in self.streams:
                main.wait_stream(s)
        self._forked = False


def concurrent_streams(dev, want, candidates=12, us=60.0):
    """`want` streams whose kernels really run side by side.  The runtime multiplexes its streams onto a few hardware
    queues (4 by default), and two streams that land on one queue serialise: four groups on what are in fact two queues
    run 30 % SLOWER than one joint graph (measured: a process that had used other streams before).  Which queue a stream
    gets is not visible through the API, so it is measured: a stream joins the set if a one-workgroup spin kernel on it
    overlaps one on every stream already chosen.  ~15 ms, once per CoGroups."""
    import os
    import time
    lib = _hip.load()
    # ONE calibration per process and device: the streams found (and the candidates) are kept, so that every plan of a run
    # sees the same answer -- a second measurement on a loaded host could classify differently, change the split of the
    # groups and with it the summation order of the trained nets -- and no call leaks a dozen streams.
    cache = _STREAM_CACHE.setdefault(str(dev), {})
    if 'cands' not in cache:
        cache['cands'] = [torch.cuda.Stream(device=dev) for _ in range(candidates)]
    while len(cache['cands']) < want:
        cache['cands'].append(torch.cuda.Stream(device=dev))
    cands = cache['cands']
    if want <= 1 or os.environ.get('MPNN_CO_CALIBRATE', '1') == '0':
        return cands[:want]
    if 'chosen' in cache and (len(cache['chosen']) >= want or cache.get('exhausted')):
        return cache['chosen'][:want]
    for c in cands:
        _hip.check(lib.mpnn_debug_noop(c.cuda_stream), 'noop')        # (the first launch on a stream binds its queue)
    torch.cuda.synchronize(dev)

    def together(a, b, links=3):
        # chains of DEPENDENT launches, enqueued alternately: two streams on one hardware queue may still overlap single
        # kernels (packets without a barrier bit run concurrently), but a dependent launch waits for everything ahead of
        # it in its queue -- which is what a training step is made of
        best = 1e9
        for _ in range(3):
            a.synchronize(); b.synchronize()
            t0 = time.perf_counter()
            for _ in range(links):
                lib.mpnn_debug_spin(1, 64, us, a.cuda_stream)
                lib.mpnn_debug_spin(1, 64, us, b.cuda_stream)
            a.synchronize(); b.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best < 1.45 * links * us * 1e-6
    chosen = [cands[0]]
    for c in cands[1:]:
        if len(chosen) == want:
            break
        if all(together(c, s) for s in chosen):
            chosen.append(c)
    cache['chosen'], cache['exhausted'] = chosen, len(chosen) < want
    return chosen


_STREAM_CACHE = {}


def _arch_signature(net):
    """What two nets must share to share launches (CoTrainer checks the launch lists themselves): the tree's
    shape and every parameter's owner type, name and shape, in link order."""
    return (tuple(len(ℓ.sinks) for ℓ in net.layers),
            tuple((type(p.owner).__name__, p.name, tuple(p.shape)) for p in net._all_params))
