"""Data-parallel training: one process per GPU, RCCL over xGMI.

The reference is single-process (scripts/train-nets:159-164); data parallelism
is new here.  Every rank holds all 680 798 parameters and its own 128-image
batch (weak scaling: BatchNorm statistics stay per replica, exactly the
reference's per-batch semantics).  The only exchange per step is ONE all-reduce
(sum) of the flat fp32 gradient buffer ``G`` -- 2.72 MB, with the per-node TALR
statistics (sum p_tr, sum p_tr^2; net_types.py:25-27) riding at its tail so the
learning-rate scales are those of the GLOBAL batch.  The optimizer kernel then
applies 1/world_size to the gradients and 1/(n*world_size) to the statistics.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT).  backend defaults to nccl (= RCCL on ROCm) when a
    GPU is visible, gloo otherwise."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return 0, 1
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def allreduce_sum(flat):
    """The one data-path collective of a step."""
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def attach(net):
    """Make ``net.train.run`` data-parallel over the initialised process group."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return net
    eng = net.engine()
    eng.world = dist.get_world_size()
    eng.allreduce = allreduce_sum
    eng._graphs.clear()                        # graphs captured so far folded the optimizer into the step
    for buf in (eng.P, eng.S, eng.A):          # identical replicas to start from
        dist.broadcast(buf, src=0)
    return net
