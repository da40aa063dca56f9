"""CPU, world_size 2 over gloo: the data-parallel reducer (lib/_dp.py).  One all-reduce (sum) of the
flat gradient buffer that also carries the per-node TALR statistics (at its head in the engine's layout; where
they sit does not matter to the reducer); the optimizer's scaling (grad/world, statistics/(n*world)) must
reproduce the single-process global-batch update."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, 'multipath-nn_amd')):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from lib import _dp
    r, w = _dp.init('gloo')
    assert (r, w) == (rank, world)
    g = np.random.default_rng(rank)
    n_params, n_nodes, n = 1000, 5, 16
    grads = g.standard_normal(n_params).astype(np.float32)          # per-replica mean-loss gradients
    p_tr = g.random((n_nodes, n)).astype(np.float32)                 # per-replica routing probabilities
    stat = np.stack([p_tr.sum(1), (p_tr ** 2).sum(1)], 1).reshape(-1)
    flat = torch.from_numpy(np.concatenate([stat, grads]))            # statistics first, as in Engine.G
    _dp.allreduce_sum(flat)
    out[rank] = (grads, p_tr, flat.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_with_talr_tail_matches_global_batch():
    world, port = 2, 29517
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    grads = np.stack([out[r][0] for r in range(world)])
    p_tr = np.concatenate([out[r][1] for r in range(world)], axis=1)            # the global batch
    n = out[0][1].shape[1]
    for r in range(world):
        flat = out[r][2]
        stat, g_sum = flat[:10].reshape(-1, 2), flat[10:]
        # optimizer scaling (mpnn_talr_momentum_step): grad_scale = 1/world, inv_n = 1/(n*world)
        assert np.allclose(g_sum / world, grads.mean(0), atol=1e-6)
        assert np.allclose(stat[:, 0] / (n * world), p_tr.mean(1), atol=1e-6)
        s_dp = 1 / np.sqrt(stat[:, 1] / (n * world))
        assert np.allclose(s_dp, 1 / np.sqrt((p_tr ** 2).mean(1)), rtol=1e-5)     # net_types.py:25-27 on the global batch


def test_single_process_is_a_noop():
    for p in (ROOT, os.path.join(ROOT, 'multipath-nn_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.pop('WORLD_SIZE', None)
    from lib import _dp
    if not dist.is_initialized():
        assert _dp.init() == (0, 1)
