"""Seeded fuzz over the shapes that select the special forward bodies of mpnn_msconv_fwd_group at evaluation-size
batches -- the first conv (conv_first.hip), the one-chunk / multi-chunk / image + V strip bodies (conv_strip.h), the
32-channel output tiles (fwd_group_k WIDE) -- against the general body (mpnn_msconv_fwd) on the same inputs:
equal to fp32 summation order (bit-identical where the contraction order is the same), the pooled map the exact
max-pool of the launch's own output, the statistics the same sums.  Which body ran is a function of the shapes and the
sample capacity alone (include/mpnn_hip.h); the test only states the contract every body has to keep."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def cases():
    rng = np.random.default_rng(20261003)
    out = []
    for k in range(14):
        big = rng.random() < 0.7
        if big:
            H = int(rng.choice([8, 12, 16, 32])); W = int(rng.choice([16, 32]))
        else:
            H = W = int(rng.choice([4, 8]))
        ca = int(rng.choice([1, 3, 16, 32])) if big else int(rng.choice([32, 64]))
        cv = int(rng.choice([0, 16, 32]))
        if ca in (1, 3) and rng.random() < 0.5:
            cv = 0
        co = int(rng.choice([16, 32, 64]))
        n = int(rng.choice([512, 517, 1024, 1031]))
        shift = int(rng.choice([0, 1])) if ca in (1, 3) and cv else 0
        mode = 'id' if ca in (1, 3) else str(rng.choice(['id', 'batch', 'moving']))
        out.append((H, W, n, ca, cv, co, shift, mode))
    # (every special body at least once, whatever the draw)
    out += [(8, 8, 1031, 32, 32, 64, 0, 'moving'), (4, 4, 1024, 64, 0, 128, 0, 'moving'), (16, 16, 512, 3, 16, 16, 1, 'id'),
            (32, 32, 512, 16, 0, 16, 0, 'moving'), (16, 32, 512, 32, 16, 32, 0, 'batch')]
    return out


@pytest.mark.parametrize('case', cases())
def test_group_bodies_keep_the_contract(case):
    import hiputil as U
    from lib import _hip
    H, W, n, ca, cv, co, shift, mode = case
    rng = np.random.default_rng(sum(int(q) * p for q, p in zip(case[:7], (3, 5, 7, 11, 13, 17, 19))))
    x = rng.standard_normal((n, H << shift, W << shift, ca)).astype(np.float32)
    v = rng.standard_normal((n, 2 * H, 2 * W, cv)).astype(np.float32) if cv else None
    wh = (rng.standard_normal((3, 3, ca, co)) / np.sqrt(9 * ca)).astype(np.float32)
    wv = (rng.standard_normal((3, 3, cv, co)) / np.sqrt(9 * cv)).astype(np.float32) if cv else None
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    bn, cnt, hmode = None, 1, _hip.ACT_IDENTITY
    if mode != 'id':
        gamma, beta = rng.uniform(0.5, 1.5, ca), rng.standard_normal(ca) * 0.3
        bn, cnt = U.bn_dict(x, gamma, beta, rng.standard_normal(ca) * 0.2, rng.uniform(0.5, 1.5, ca))
        hmode = _hip.ACT_BN_BATCH if mode == 'batch' else _hip.ACT_BN_MOVING
    want_pool = H >= 8 and H % 2 == 0 and W % 2 == 0
    one = U.conv_fwd(x, wh, b, v, wv, bn, hmode, shift, cnt, want_pool=want_pool)
    grp = U.conv_fwd(x, wh, b, v, wv, bn, hmode, shift, cnt, want_pool=want_pool, group=True)
    scale = 1 + np.abs(one[0]).max()
    assert np.abs(grp[0] - one[0]).max() <= 2e-5 * scale, case
    if want_pool:
        assert np.array_equal(grp[2], U.pool2_np(grp[0])), case
    ref1 = np.asarray(one[1], np.float64)
    assert np.abs(np.asarray(grp[1], np.float64) - ref1).max() <= 1e-4 * (1 + np.abs(ref1).max()), case
