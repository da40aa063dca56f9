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

The engine is assembled from pieces (one module each): Allocation (lib/_eng_alloc.py: tree classification, flat buffers),
Planner (lib/_eng_planner.py: the training program), EvalPrograms (lib/_eng_eval.py: dense and routed evaluation), Runner
(lib/_eng_run.py: staging, eager launches, hipGraph capture / replay, the interface other modules use),
DataParallelSections (lib/_eng_dp.py), KStepGraphs (lib/_eng_ksteps.py: K steps per graph) and Inspection
(lib/_eng_inspect.py: timings, result views, statistics).
"""
import os

import torch

from lib import _hip
from lib._eng_common import (BLOCK_COMPS, CAPTURE_MODE, HEAD_COMPS, OPT_CHUNK, ROUTER_COMPS, BoundInput, _attr, _Block, _kind, _nf,   # noqa: F401
                             _Node)
from lib._eng_alloc import Allocation
from lib._eng_planner import Planner
from lib._eng_eval import EvalPrograms
from lib._eng_run import Runner
from lib._eng_dp import DataParallelSections
from lib._eng_ksteps import KStepGraphs
from lib._eng_inspect import Inspection


class Engine(Allocation, Planner, EvalPrograms, Runner, DataParallelSections, KStepGraphs, Inspection):
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
