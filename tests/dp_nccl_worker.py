"""Worker of tests/test_dp_gpu.py::test_two_gpus_over_rccl_torchrun (launched by torchrun, one
process per GPU): three data-parallel training steps over RCCL, then every rank checks that all
replicas hold bit-identical parameters and moving averages after sync_state."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'multipath-nn_amd')):
    sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist


def main():
    import arch_and_hypers as A
    from lib import _dp
    rank, world = _dp.init('nccl')
    net = A.ac_chain(k_cpt=1.6e-8, seed=21 + rank)((32, 32, 3), (10,))     # different seeds: attach() must broadcast rank 0's
    net.to('cuda:%d' % int(os.environ['LOCAL_RANK']))
    _dp.attach(net)
    eng = net.engine()
    g = np.random.default_rng(100 + rank)
    x0 = g.random((128, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[g.integers(0, 10, 128)]
    for t in range(4):
        net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.τ: 1.0})
    _dp.sync_state(net)
    torch.cuda.synchronize()
    for buf in (eng.P, eng.A, eng.S):
        mine = buf.clone()
        ref = buf.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(mine, ref), 'replicas diverged on rank %d' % rank
    assert torch.isfinite(eng.P).all()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print('dp_nccl_worker ok: %d ranks, replicas bit-identical' % world)


if __name__ == '__main__':
    main()
