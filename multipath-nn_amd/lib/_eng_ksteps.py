"""K training steps as ONE hipGraph replay (`net.train.run_steps`): per-step schedule values and per-sample k_cpt vectors in
device rings, the input pipeline's record slots, and -- under data parallelism -- the K steps' captured all-reduces."""
import ctypes as C
import os

import numpy as np
import torch

from lib import _hip
from lib.net_types import n_leaves, params_list_rec
from lib._eng_common import (BLOCK_COMPS, CAPTURE_MODE, HEAD_COMPS, OPT_CHUNK, ROUTER_COMPS, BoundInput, _attr, _Block, _kind, _nf,
                             _Node)


class KStepGraphs:

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
