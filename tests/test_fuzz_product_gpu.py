"""GPU: the HIP path on RANDOMLY DRAWN nets and hyper-parameter sets -- the draws of tests/golden/fuzz_ref_graph.py (every
key of ActorNet / CriticNet.default_hypers; chains of two to four blocks and 2- / 3-way forks; per-sample k_cpt), which
the CPU suite holds the oracle to through the reference's own graph code (tests/test_fuzz_ref_graph.py) -- against that
oracle, decision-forced, at the whole-net tolerances (tests/test_net_parity.py::run_case: every gradient and update within
1e-4 of the tensor's scale, p_ev / delta_cor exact, statistics within 1e-3).  Closes the loop reference code -> oracle ->
kernels on hyper-parameter corners no fixed case has (alpha_rtr != 1 with and without TALR, k_dec = 0, large epsilon,
mu_lrn = 0, ...)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import fuzz_ref_graph as Z
from test_net_parity import run_case

pytestmark = pytest.mark.gpu

SEEDS = [int(s) for s in os.environ.get('MPNN_FUZZ_GPU_SEEDS', '0 1 2 3 5 8 13 21 34 55 89 100 101 102').split()]


@pytest.mark.parametrize('seed', SEEDS)
def test_product_on_fuzzed_hypers(seed):
    import arch_and_hypers as A
    import lib.net_types as NT
    case = Z.draw_case(seed)
    tau = case['tau'] if case['tau'] is not None else case['hypers'].get('τ')
    n = 8 + 4 * (seed % 5)
    rng = np.random.RandomState(500 + seed)
    kv = (lambda t, n_: np.asarray(Z.K_CPTS, np.float32)[rng.randint(0, len(Z.K_CPTS), n_)]) if case['dyn'] else None
    memo = {}

    def kv_memo(t, n_):                       # (run_case asks twice per step: the same vector both times)
        if (t, n_) not in memo:
            memo[(t, n_)] = kv(t, n_)
        return memo[(t, n_)]
    feeds = (lambda net, t: {}) if case['kind'] == 'SRNet' else (lambda net, t: {net.τ: tau})
    print('seed %d: %s %s %s' % (seed, case['kind'], case['shape'], {k: v for k, v in case['hypers'].items()}))
    run_case(lambda x0s, ys: Z.build(A, NT, case), n, feeds, steps=2, k_cpt_vec=kv_memo if kv else None)
