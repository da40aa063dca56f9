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


def init(backend=None, force=False):
    """Initialise torch.distributed from the torchrun environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT).  backend defaults to nccl (= RCCL on ROCm) when a
    GPU is visible, gloo otherwise.  force: create the process group even for one rank
    (tests exercise the RCCL code path on a single GPU that way)."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1 and not force:
        return 0, 1
    if force:
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
    if backend is None:
        backend = os.environ.get('MPNN_DP_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(local_device())
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def local_device():
    """GPU of this rank: LOCAL_RANK, or 0 for every rank with MPNN_DP_ONE_GPU=1 (functional tests of the
    multi-rank paths on a one-GPU box, together with MPNN_DP_BACKEND=gloo: RCCL refuses two ranks per GPU)."""
    return 0 if os.environ.get('MPNN_DP_ONE_GPU') else int(os.environ.get('LOCAL_RANK', '0'))


def allreduce_sum(flat):
    """Blocking all-reduce (sum) of a flat tensor."""
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_async(flat):
    """All-reduce (sum) of one gradient bucket, asynchronous: with RCCL the collective is queued on
    the process group's own stream behind everything already on the compute stream, the caller keeps
    queueing the rest of the backward pass, and ``handle.wait()`` later makes the compute stream wait
    for it (a stream dependency, the host does not block).  xGMI is point to point (7 links of
    ~153 GB/s per GPU) and a bucket is 0.3-1.3 MB, so each all-reduce is latency- not bandwidth-bound:
    three buckets, two of them hidden behind the remaining backward pass."""
    if flat.is_cuda and dist.get_backend() == 'gloo':          # (test configuration: gloo without device support)
        h = flat.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        flat.copy_(h)
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_sum_any(flat):
    """Blocking form of allreduce_async (bench: the collective timed by itself)."""
    h = allreduce_async(flat)
    if h is not None:
        h.wait()
    return flat


def sync_state(net):
    """BatchNorm moving averages are per-replica state (every rank normalises with the statistics of
    its own 128 images, the reference's per-batch semantics, and its averages drift apart by sampling
    noise).  Before anything that READS them -- the statistics pass, a checkpoint -- the ranks agree on
    their mean, so every rank evaluates the same function and rank 0's file is not one replica's view."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return net
    eng = net.engine()
    allreduce_sum_any(eng.S)
    eng.S.div_(dist.get_world_size())
    return net


def attach(net, force=False):
    """Make ``net.train.run`` data-parallel over the initialised process group."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return net
    eng = net.engine()
    if not hasattr(eng, 'dp_buckets'):
        raise NotImplementedError('%s has no gradient buckets: data-parallel training covers the multiscale chain / tree '
                                  'engine (lib/_plan.py), not the single-scale Conv engine' % type(eng).__name__)
    eng.world = dist.get_world_size()
    eng.allreduce = allreduce_async
    eng.allreduce_capturable = dist.get_backend() == 'nccl' and eng.P.is_cuda     # RCCL collectives capture into hipGraphs
    eng._graphs.clear()                        # graphs captured so far folded the optimizer into the step
    for buf in (eng.P, eng.S, eng.A):          # identical replicas to start from
        if buf.is_cuda and dist.get_backend() == 'gloo':
            h = buf.cpu(); dist.broadcast(h, src=0); buf.copy_(h)
        else:
            dist.broadcast(buf, src=0)
    eng.invalidate_packs()
    return net


def detach(net):
    """Back to single-process training (bench: after the 1-rank structure measurement)."""
    eng = net.engine()
    eng.world, eng.allreduce, eng.allreduce_capturable = 1, None, False
    eng._graphs.clear()
    return net
