"""GPU: the multi-stream DAG schedule and hipGraph replay give the same step as the
sequential eager order (same kernels, same inputs; only atomics ordering may differ)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run(multi, graph, steps=4):
    import arch_and_hypers as A
    net = A.ac_chain(k_cpt=1.6e-8, seed=7)((32, 32, 3), (10,))
    eng = net.engine()
    eng.multi_stream, eng.use_graph = multi, graph
    rng = np.random.default_rng(3)
    x0 = rng.random((32, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 32)]
    for t in range(steps):
        net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.τ: 1.0})
    torch.cuda.synchronize()
    return eng.P.cpu().numpy().copy(), eng.S.cpu().numpy().copy()


def test_multi_stream_and_graph_match_sequential():
    p0, s0 = run(False, False)
    for multi, graph in ((True, False), (False, True), (True, True)):
        p1, s1 = run(multi, graph)
        assert np.abs(p1 - p0).max() <= 1e-4 * np.abs(p0).max(), (multi, graph)
        assert np.abs(s1 - s0).max() <= 1e-4 * np.abs(s0).max(), (multi, graph)
