"""What the pieces of the engine (lib/_plan.py and its _eng_* modules) share: the kernel library binding, the chain shapes
the planner accepts, attribute lookup through Python's identifier normalisation, and the small record classes."""
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


