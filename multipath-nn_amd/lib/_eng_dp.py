"""The data-parallel step: the program cut into sections at its gradient-bucket points, the asynchronous all-reduce of a
bucket (lib/_dp.py installs `allreduce`), the optional per-bucket update on a side stream."""
import ctypes as C
import os

import numpy as np
import torch

from lib import _hip
from lib.net_types import n_leaves, params_list_rec
from lib._eng_common import (BLOCK_COMPS, CAPTURE_MODE, HEAD_COMPS, OPT_CHUNK, ROUTER_COMPS, BoundInput, _attr, _Block, _kind, _nf,
                             _Node)


class DataParallelSections:

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


    @property
    def per_bucket_update(self):
        """Data parallel with several gradient buckets: each bucket is applied on a side stream behind its own all-reduce."""
        return bool(self._bucket_opt_on())

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
