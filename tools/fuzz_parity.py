"""Parity fuzz beyond the fixed test cases (GPU): the exit-path affine maps over random batch sizes /
channel counts, and one whole training step of the actor and critic chains against the decision-forced
oracle over ragged batch sizes.

    python tools/fuzz_parity.py [n_lin_cases]

Known and expected: batch size 2 misses the 1e-4 gradient tolerance by a factor of ~2 on a few conv / BatchNorm
tensors.  With two samples the 1x1 maps of the deepest blocks give BatchNorm exactly two values per channel:
its output is +-gamma + beta whatever the input, the true input gradient is zero, and what is compared is
fp32 rounding noise scaled by rstd = 2 / |a - b|.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import arch_and_hypers as A
import test_exit_kernels as TE
import test_net_parity as TN

n_lin = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(0)
bad = 0
for k in range(n_lin):
    n = int(rng.integers(1, 300)); C_ = int(rng.choice([16, 32, 64, 128])); dyn = bool(rng.integers(0, 2))
    try:
        TE.test_lin_fwd_bwd(n, C_, dyn)
    except AssertionError as e:
        bad += 1; print('FAIL lin', n, C_, dyn, str(e)[:200])
print('lin_fwd / lin_fwd_ks / lin_bwd / lin_bwd_rs: %d cases, %d failures' % (n_lin, bad))
bad = 0
sizes = (1, 3, 17, 33, 50, 64, 65, 99, 127)
for n in sizes:
    for mk, name in ((lambda: A.ac_chain(k_cpt=1.6e-8), 'ac_chain'), (lambda: A.cr_chain(k_cpt=8e-9), 'cr_chain')):
        try:
            TN.run_case(mk(), n, lambda net, t: {net.τ: 0.7}, steps=1, tol=2e-4)
        except AssertionError as e:
            bad += 1; print('FAIL', name, n, str(e)[:300])
print('one training step vs the decision-forced oracle: %d cases, %d failures' % (2 * len(sizes), bad))
