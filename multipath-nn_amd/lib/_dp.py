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

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # (dmabuf IPC: what RCCL between processes needs on this stack; before HIP starts)

import torch
import torch.distributed as dist


_selftest = None          # outcome of captured_collectives_work() for the current process group


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
        # A cap on RCCL's channels (= workgroups of its kernels) for the bucketed form, where the backward launches
        # that run beside a bucket's all-reduce leave MPNN_DP_RESERVE_CUS compute units free (lib/_plan.py,
        # mpnn_set_reserved_cus) and the collective must fit into them.  The default form (one all-reduce of 2.7 MB
        # after the backward pass, nothing running beside it) leaves RCCL its own choice.
        if int(os.environ.get('MPNN_DP_RESERVE_CUS', '0')) > 0:
            os.environ.setdefault('NCCL_MAX_NCHANNELS', os.environ['MPNN_DP_RESERVE_CUS'])
    global _selftest
    _selftest = None
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


def max_divergence(t):
    """max over ranks and elements of |t - rank 0's t| (a float; 0.0 for identical replicas).  Diagnostics of a
    multi-rank run (bench.py): the replicas' parameters must still agree bit for bit after any number of steps."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0.0
    staged = t.is_cuda and dist.get_backend() == 'gloo'
    ref = t.detach().cpu().clone() if staged else t.detach().clone()
    dist.broadcast(ref, src=0)
    d = (t.detach().to(ref.device) - ref).abs().max().reshape(1).double()
    dist.all_reduce(d, op=dist.ReduceOp.MAX)
    return float(d.item())


def quiesce():
    """Call before capturing a hipGraph that contains collectives.  The process group's watchdog thread polls the end
    event of every collective issued OUTSIDE a capture until it has seen it complete (one pass every ~100 ms); an
    event query that lands while the collective's stream is being captured fails with hipErrorCapturedEvent and takes
    the process down (seen on this stack: 'operation not permitted on an event last recorded in a capturing stream' from
    the watchdog).  After a device synchronize every pending collective is complete; two watchdog periods later the
    watchdog has dropped them all and has nothing left to poll."""
    import time
    if dist.is_initialized() and dist.get_backend() == 'nccl':
        torch.cuda.synchronize()
        time.sleep(0.3)


def agree(ok):
    """Logical AND of a local flag over the ranks (every rank must take the same form of the step: a rank whose
    graph capture failed issues its collectives from the host, the others inside their graphs)."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32,
                     device=('cuda:%d' % local_device()) if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def captured_collectives_work():
    """Self-test, run once per process group with MORE than one rank before any training step is captured: two dependent
    all-reduces recorded into ONE hipGraph (as K training steps per graph hold K of them) and replayed twice must give the sums
    over the ranks, on every rank.  The
    one-graph form of the data-parallel step (lib/_plan.py) is only used when this passed everywhere; otherwise all
    ranks use one graph per bucket section with the collectives issued from the host."""
    import warnings
    dev = 'cuda:%d' % local_device()
    rank, world = dist.get_rank(), dist.get_world_size()
    buf = torch.zeros(1024, device=dev)
    g = None
    try:
        src = torch.full((1024,), float(rank + 1), device=dev)
        side = torch.cuda.Stream(device=dev)
        dist.all_reduce(buf.clone(), op=dist.ReduceOp.SUM)        # (communicator set-up outside the capture)
        quiesce()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            buf.copy_(src)
            for _ in range(2):                                    # (TWO dependent collectives in one graph: the K-step form)
                h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
                with torch.cuda.stream(side):                     # (the shape of the real step: a consumer on a third stream)
                    h.wait()
                    buf.mul_(2.0)
                torch.cuda.current_stream().wait_stream(side)
    except Exception as e:                                          # noqa: BLE001 -- any failure means "do not capture"
        warnings.warn('capturing an RCCL collective failed (%r): the data-parallel step uses section graphs' % (e,))
        g = None
    # The ranks agree that EVERY rank captured before ANY rank replays: a rank whose capture threw would otherwise sit in
    # the eager MIN all-reduce below while the others replay a captured SUM all-reduce -- mismatched collectives, a hang
    # until the watchdog's timeout, which is the failure this self-test exists to rule out.
    torch.cuda.synchronize()
    if not agree(g is not None):
        return False
    ok = True
    try:
        quiesce()
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        ok = bool(torch.all(buf == float(2 * world * world * (world + 1))).item())      # sum(rank + 1) x 2, summed again, x 2
    except Exception as e:                                          # noqa: BLE001
        warnings.warn('replaying a captured RCCL collective failed (%r): the data-parallel step uses section graphs' % (e,))
        ok = False
    return agree(ok)


def attach(net, force=False):
    """Make ``net.train.run`` data-parallel over the initialised process group."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return net
    eng = net.engine()
    if not hasattr(eng, 'dp_buckets'):
        # the single-scale Conv engine (lib/_plan_conv.py: eager launches, no BatchNorm): one blocking all-reduce of G in
        # front of its optimizer launch, identical replicas to start from
        eng.world, eng.allreduce = dist.get_world_size(), allreduce_async
        for buf in (eng.P, eng.A):
            if buf.is_cuda and dist.get_backend() == 'gloo':
                h = buf.cpu(); dist.broadcast(h, src=0); buf.copy_(h)
            else:
                dist.broadcast(buf, src=0)
        return net
    eng.world = dist.get_world_size()
    eng.allreduce = allreduce_async
    eng.allreduce_capturable = dist.get_backend() == 'nccl' and eng.P.is_cuda     # RCCL collectives capture into hipGraphs
    eng.dp_agree = agree
    eng.dp_quiesce = quiesce
    if eng.allreduce_capturable and eng.dp_one_graph and eng.world > 1 and eng.use_graph:
        # the one-graph step has only ever run with a forced ONE-rank group on the pool's one-GPU boxes: with more
        # ranks it is taken only if a captured all-reduce verifiably works on this stack, and every rank agrees
        global _selftest
        if _selftest is None:
            _selftest = captured_collectives_work()
        eng.allreduce_capturable = _selftest
        eng.dp_selftest = bool(_selftest)
    eng.drop_graphs()                        # graphs captured so far folded the optimizer into the step
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
    if not hasattr(eng, 'dp_buckets'):
        eng.world, eng.allreduce = 1, None
        return net
    eng.world, eng.allreduce, eng.allreduce_capturable, eng.dp_agree, eng.dp_quiesce = 1, None, False, None, None
    eng.drop_graphs()
    return net
