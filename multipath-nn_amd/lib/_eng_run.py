"""Running a program: staging a feed (inputs, schedule values through a ring of pinned buffers), eager launches or hipGraph
capture and replay of a whole step, the clearing discipline of the step's accumulators, the optimizer launch, the
input-pipeline prologue -- plus the small public interface other modules drive an engine through (lib/_co.py,
lib/data.py, the drivers)."""
import ctypes as C
import os

import numpy as np
import torch

from lib import _hip
from lib.net_types import n_leaves, params_list_rec
from lib._eng_common import (BLOCK_COMPS, CAPTURE_MODE, HEAD_COMPS, OPT_CHUNK, ROUTER_COMPS, BoundInput, _attr, _Block, _kind, _nf,
                             _Node)


class Runner:


    # ------------------------------------------------------------------ the interface other modules drive an engine through
    # (lib/_co.py steps several engines in one launch list; lib/data.py installs the input pipeline; the drivers size the
    # buffers.  Everything they need is here -- none of them reaches into the planner's or the graph cache's state.)
    def ensure_capacity(self, n, train=True):
        """Size the activation / gradient buffers for batches of n (reallocation drops the programs and graphs that baked
        the old pointers in; `generation` changes)."""
        return self._ensure_capacity(n, train)

    @property
    def generation(self):
        """Changes whenever the engine reallocated buffers that programs, graphs or argument tables of OTHER modules may
        have captured (a larger batch came by)."""
        return getattr(self, '_gen', 0)

    def drop_programs(self):
        """Forget the cached programs and graphs (a setting that selects them -- `hyp`, `co_share`, the prologue -- changed)."""
        self._progs.clear()
        self._graphs.clear()

    def drop_graphs(self):
        """Forget the captured graphs, keep the programs (the step's FORM changed: an all-reduce was attached or detached)."""
        self._graphs.clear()

    def step_graph_form(self):
        """How the data-parallel training step is replayed, once captured: 'whole' (the step with its collectives is ONE
        hipGraph), 'sections' (one graph per gradient-bucket section, collectives issued from the host) or None."""
        for k, g in self._graphs.items():
            if k[0] == 'tr' and len(k) > 2 and k[2] and isinstance(g, tuple):
                return 'whole' if g[1] == 'whole' else 'sections'
        return None

    def k_step_graph_captured(self, dp=None):
        """A K-step training graph (run_steps) has been captured (dp: only data-parallel / only single-process ones)."""
        return any(k[0] == 'trK' and not isinstance(g, str) and (dp is None or bool(k[-1]) == bool(dp)) for k, g in self._graphs.items())

    def stage_feed(self, feed):
        """Stage a feed's inputs WITHOUT uploading its schedule values: returns (n, mode, values) -- the caller uploads the
        values of several engines at once (co-training: one [K, MPNN_HYP_N] buffer whose rows are the engines' `hyp`)."""
        n, mode = self._stage(feed, upload_hyp=False)
        return n, mode, self._hyp_stage

    def schedule_values(self, feed, n):
        """The MPNN_HYP_N schedule / hyper-parameter values of one step as a host tensor (a fresh copy)."""
        return self._hyp_values(feed, n).clone()

    def hyp_rewritten(self):
        """Somebody else rewrote this engine's device `hyp` row (a joint graph's per-step copy): the next solo step uploads
        its values whatever it sent last."""
        self._hyp_sent = None
        self._hyp_epoch = getattr(self, '_hyp_epoch', 0) + 1

    @property
    def hyp_epoch(self):
        """Counts the uploads / rewrites of this engine's device `hyp` row (a co-trainer re-sends its buffer when a net
        stepped alone in between)."""
        return getattr(self, '_hyp_epoch', 0)

    def fresh_packs(self):
        """Re-pack the weights if the parameters were written from outside a training step (eagerly: never inside a capture)."""
        if not self._packs_fresh:
            self._pack()
            self._packs_fresh = True

    @property
    def accumulators_clean(self):
        """The step's accumulators (BatchNorm slot sums, TALR statistics, loss sums) are cleared: a program that folds the
        clearing into its last readers may start without a clearing launch."""
        return self._acc_clean

    def begin_step(self, fold):
        """In front of a training step launched from OUTSIDE run(): a clearing launch unless the previous step left the
        accumulators cleared and this program relies on that (fold); marks them dirty until `end_step`."""
        if not (fold and self._acc_clean):
            self._begin(True)
        self._acc_clean = False

    def clear_for_capture(self):
        """Before a fold program is CAPTURED (the graph holds no clearing launch): start from cleared accumulators."""
        if not self._acc_clean:
            self._begin(True)
            self._acc_clean = True

    def end_step(self, n, fold):
        """Behind a training step launched from outside run(): what the step left behind, and the result views."""
        self._acc_clean = bool(fold)
        self.last_n, self.last_mode, self._last_fold = n, 'tr', bool(fold)
        self._bind_views(n)

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
