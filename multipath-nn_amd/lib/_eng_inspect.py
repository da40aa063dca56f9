"""Looking at an engine: per-launch / per-family HIP-event timings of a program, the views that bind results to the layer
objects (`p_tr`, `p_ev`, `c_err`, ...), and the statistics tensors of the harness (`state()`)."""
import ctypes as C
import os

import numpy as np
import torch

from lib import _hip
from lib.net_types import n_leaves, params_list_rec
from lib._eng_common import (BLOCK_COMPS, CAPTURE_MODE, HEAD_COMPS, OPT_CHUNK, ROUTER_COMPS, BoundInput, _attr, _Block, _kind, _nf,
                             _Node)


class Inspection:

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


    def state_sums(self):
        """The statistics of state(), each SUMMED over the samples of the last run (float64 device tensors, same keys): what
        the dataset-wide averages of lib/desc.py need.  A dozen batched device operations (row gathers of the flat result
        buffers, two small matrix products for the per-class statistics) instead of six per leaf -- a statistics pass spent
        as long in ~150 tiny launches per batch as in the evaluation itself (profiles/r06_experiment_wall.txt)."""
        net, n = self.net, self.last_n
        nn, nl, MS, dev = len(self.nodes), len(self.leaves), self.max_sinks, self.dev
        c = getattr(self, '_sums_cache', None)
        if c is None:
            c = self._sums_cache = dict(
                leaf_rows=torch.tensor([nd.idx for nd in self.leaves], device=dev),
                leaf_ids=torch.tensor([nd.leaf_id for nd in self.leaves], device=dev),
                ops=torch.tensor(self.node_ops_host, dtype=torch.float64, device=dev),
                sw_ids=torch.tensor([nd.switch_id for nd in self.switches], device=dev, dtype=torch.long),
                sw_mask=torch.tensor([[1.0 / len(nd.layer.sinks) if k < len(nd.layer.sinks) else 0.0 for k in range(MS)]
                                      for nd in self.switches], dtype=torch.float64, device=dev))
        pev = self.p_ev[:nn * n].view(nn, n).double()
        pl = pev[c['leaf_rows']]                                   # [leaves, n], in self.leaves order
        dcor = self.d_cor[:nl * n].view(nl, n)[c['leaf_ids']].double()
        cerr = self.c_err[:nl * n].view(nl, n)[c['leaf_ids']].double()
        y = self.y[:n].double()
        cor = pl * dcor
        inc = pl - cor                                             # (δ_cor is 0 or 1: p_ev (1 - δ) exactly)
        cor_cls, inc_cls = cor @ y, inc @ y
        p_cor, p_inc, c_sum = cor.sum(1), inc.sum(1), cerr.sum(1)
        out = {(net, 'acc'): p_cor.sum(), (net, 'moc'): (pev.sum(1) * c['ops']).sum()}
        ptr = self.p_tr[:nn * n].view(nn, n)[c['leaf_rows']].double().sum(1) if net._net_kind != 'sr' else None
        for k, nd in enumerate(self.leaves):
            ℓ = nd.layer
            out[(ℓ, 'p_cor')], out[(ℓ, 'p_inc')] = p_cor[k], p_inc[k]
            out[(ℓ, 'p_cor_by_cls')], out[(ℓ, 'p_inc_by_cls')] = cor_cls[k], inc_cls[k]
            if ptr is not None:
                out[(ℓ, 'p_tr')] = ptr[k]
            out[(ℓ, 'c_err')] = c_sum[k]
        if self.switches:
            nsw = len(self.switches)                                                      # (switch ids: 0 .. nsw - 1)
            r = self.r[:nsw * n * MS].view(nsw, n, MS)[c['sw_ids']].abs().double()        # [switches, n, MS]
            x = (r * c['sw_mask'][:, None, :]).sum((1, 2))                                  # sum over samples of mean |r| over the sinks
            for k, nd in enumerate(self.switches):
                out[(nd.layer, 'x_rte')] = x[k]
        return out

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
