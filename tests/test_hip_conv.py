"""GPU parity: the conv-path HIP kernels (through the C ABI) vs. the float64 oracle.

Every convolution instance of the shipped architecture is covered,
plus ragged batches (partial 4-image tiles on the 4x4 maps).  Tolerances: fp32
MFMA is an exact fp32 fma chain, so errors are rounding only:
  forward / dgrad  : max|err| <= 2e-5 * (1 + max|ref|)
  wgrad            : max|err| <= 1e-4 * (1 + max|ref|)   (long fp32 sums + atomics)
"""
import numpy as np
import pytest

from oracle import np_ops as O

pytestmark = pytest.mark.gpu

# (H, Ca, Cv, Cout, shift) -- every conv of arch (scripts/arch_and_hypers.py:19-27)
SHAPES = [
    (32, 3, 0, 16, 0), (16, 3, 16, 16, 1), (8, 3, 16, 16, 2), (4, 3, 16, 16, 3),       # L0
    (32, 16, 0, 16, 0), (16, 16, 16, 16, 0), (8, 16, 16, 16, 0), (4, 16, 16, 16, 0),   # L1
    (16, 16, 0, 32, 0), (8, 16, 32, 32, 0), (4, 16, 32, 32, 0),                        # L2
    (16, 32, 0, 32, 0), (8, 32, 32, 32, 0), (4, 32, 32, 32, 0),                        # L3
    (8, 32, 0, 64, 0), (4, 32, 64, 64, 0),                                             # L4
    (8, 64, 0, 64, 0), (4, 64, 64, 64, 0),                                             # L5
    (4, 64, 0, 128, 0), (4, 128, 0, 128, 0),                                           # L6, L7
    (32, 1, 0, 16, 0), (8, 1, 16, 16, 2),                                              # MNIST variant
    (16, 32, 32, 64, 0), (32, 16, 0, 64, 0),      # beyond the shipped specs: 64-channel tiles on W >= 16 maps
]


def close(got, ref, tol):
    ref = np.asarray(ref, np.float64)
    err = np.abs(np.asarray(got, np.float64) - ref).max()
    assert err <= tol * (1.0 + np.abs(ref).max()), (err, np.abs(ref).max())


def make(shape, rng, n):
    H, ca, cv, co, shift = shape
    x = rng.standard_normal((n, H << shift, H << shift, ca))
    v = rng.standard_normal((n, 2 * H, 2 * H, cv)) if cv else None
    wh = rng.standard_normal((3, 3, ca, co)) / 3 / np.sqrt(ca)
    wv = rng.standard_normal((3, 3, cv, co)) / 3 / np.sqrt(cv) if cv else None
    b = rng.standard_normal(co) * 0.1
    return x, v, wh, wv, b


def ref_fwd(x, v, wh, wv, b, shift, bn=None):
    xs = x[:, ::1 << shift, ::1 << shift, :]
    if bn is not None:
        gamma, beta = bn
        y, _, _ = O.bn_train(xs, gamma, beta)
        xs = O.relu(y)
    out = b + O.conv_same(xs, wh)
    if v is not None:
        out = out + O.conv_same(O.pool2(v), wv)
    return out, xs


@pytest.mark.parametrize('shape', SHAPES)
def test_conv_fwd_raw_and_bn(shape):
    import hiputil as U
    from lib import _hip
    H, ca, cv, co, shift = shape
    rng = np.random.default_rng(hash(shape) % 2 ** 31)
    n = 5 if H == 4 else 2
    x, v, wh, wv, b = make(shape, rng, n)
    # identity input (pyramid); the epilogue also max-pools the output for the next coarser scale
    if H >= 8:
        out, osum, pooled = U.conv_fwd(x, wh, b, v, wv, shift=shift, want_pool=True)
        assert np.array_equal(pooled, U.pool2_np(out))
    else:
        out, osum = U.conv_fwd(x, wh, b, v, wv, shift=shift)
    ref, _ = ref_fwd(x, v, wh, wv, b, shift)
    close(out, ref, 2e-5)
    close(osum[:co], ref.sum((0, 1, 2)), 1e-5)
    close(osum[co:], (ref ** 2).sum((0, 1, 2)), 1e-5)
    if shift == 0 and ca >= 4:
        # BatchNorm + ReLU applied on load, batch statistics
        gamma, beta = rng.uniform(0.5, 1.5, ca), rng.standard_normal(ca) * 0.3
        bn, cnt = U.bn_dict(x, gamma, beta)
        out, _ = U.conv_fwd(x, wh, b, v, wv, bn=bn, mode=_hip.ACT_BN_BATCH, bn_cnt=cnt)
        ref, _ = ref_fwd(x, v, wh, wv, b, 0, bn=(gamma, beta))
        close(out, ref, 3e-5)
        # moving-average mode
        m_avg, v_avg = rng.standard_normal(ca) * 0.2, rng.uniform(0.5, 2.0, ca)
        bn, cnt = U.bn_dict(x, gamma, beta, m_avg, v_avg)
        out, _ = U.conv_fwd(x, wh, b, v, wv, bn=bn, mode=_hip.ACT_BN_MOVING, bn_cnt=cnt)
        xs = O.relu(O.bn_eval(x, gamma, beta, m_avg, v_avg))
        ref = b + O.conv_same(xs, wh) + (O.conv_same(O.pool2(v), wv) if v is not None else 0)
        close(out, ref, 3e-5)


@pytest.mark.parametrize('shape', [s for s in SHAPES if s[1] >= 16])
def test_dgrad_horz(shape):
    import hiputil as U
    H, ca, cv, co, _ = shape
    rng = np.random.default_rng(1 + hash(shape) % 2 ** 31)
    n = 5 if H == 4 else 2
    g = rng.standard_normal((n, H, H, co))
    w = rng.standard_normal((3, 3, ca, co)) / 3 / np.sqrt(ca)
    dy_ref, _ = O.conv_same_bwd(np.zeros((n, H, H, ca)), w, g)
    out, _ = U.dgrad_horz(g, w)
    close(out, dy_ref, 2e-5)
    # fused producer BatchNorm+ReLU backward reductions (+ extra gradient)
    s = rng.standard_normal((n, H, H, ca))
    extra = rng.standard_normal((n, H, H, ca))
    gamma, beta = rng.uniform(0.5, 1.5, ca), rng.standard_normal(ca) * 0.3
    bn, cnt = U.bn_dict(s, gamma, beta)
    out, red = U.dgrad_horz(g, w, s_prev=s, bn=bn, cnt=cnt, extra=extra)
    y, m, var = O.bn_train(s, gamma, beta)
    dz = np.where(y > 0, dy_ref + extra, 0.0)
    xh = (s - m) / np.sqrt(var + 1e-6)
    # the ReLU mask may flip for |y| ~ 1e-7: compare away from the kink
    safe = np.abs(y) > 1e-4
    assert np.abs(out - dz)[safe].max() <= 2e-5 * (1 + np.abs(dz).max())
    close(red[:ca], dz.sum((0, 1, 2)), 1e-3)
    close(red[ca:], (dz * xh).sum((0, 1, 2)), 1e-3)


@pytest.mark.parametrize('shape', [s for s in SHAPES if s[2] >= 16])
def test_dgrad_vert(shape):
    import hiputil as U
    H, ca, cv, co, _ = shape
    rng = np.random.default_rng(2 + hash(shape) % 2 ** 31)
    n = 5 if H == 4 else 2
    g = rng.standard_normal((n, H, H, co))
    w = rng.standard_normal((3, 3, cv, co)) / 3 / np.sqrt(cv)
    s_f = rng.standard_normal((n, 2 * H, 2 * H, cv))
    dz_f = rng.standard_normal((n, 2 * H, 2 * H, cv))
    gamma, beta = rng.uniform(0.5, 1.5, cv), rng.standard_normal(cv) * 0.3
    bn, cnt = U.bn_dict(s_f, gamma, beta)
    _, m, var = O.bn_train(s_f, gamma, beta)
    xh = (s_f - m) / np.sqrt(var + 1e-6)
    red = np.concatenate([dz_f.sum((0, 1, 2)), (dz_f * xh).sum((0, 1, 2))])
    dp, _ = O.conv_same_bwd(np.zeros((n, H, H, cv)), w, g)
    pool_g = O.pool2_bwd(s_f, dp)
    ref = gamma / np.sqrt(var + 1e-6) * (dz_f - red[:cv] / cnt - xh * red[cv:] / cnt) + pool_g
    got = U.dgrad_vert(g, w, s_f, bn, cnt, dz_fine=dz_f, red=red)
    close(got, ref, 3e-5)
    # finer BN output without consumers: g_fine = maxpool_bwd(dv) only
    got = U.dgrad_vert(g, w, s_f, bn, cnt)
    close(got, pool_g, 3e-5)


@pytest.mark.parametrize('shape', SHAPES)
def test_wgrad(shape):
    import hiputil as U
    from lib import _hip
    H, ca, cv, co, shift = shape
    rng = np.random.default_rng(3 + hash(shape) % 2 ** 31)
    n = 6 if H == 4 else 3
    x, v, wh, wv, b = make(shape, rng, n)
    g = rng.standard_normal((n, H, H, co))
    bnp = None
    if shift == 0 and ca >= 4:
        gamma, beta = rng.uniform(0.5, 1.5, ca), rng.standard_normal(ca) * 0.3
        bn, cnt = U.bn_dict(x, gamma, beta)
        dwa, dwv, db = U.wgrad(x, g, v, bn=bn, mode=_hip.ACT_BN_BATCH, bn_cnt=cnt)
        bnp = (gamma, beta)
    else:
        dwa, dwv, db = U.wgrad(x, g, v, shift=shift, n_split=1)      # direct write, no slab
    _, xs = ref_fwd(x, v, wh, wv, b, shift, bn=bnp)
    _, dwa_ref = O.conv_same_bwd(xs, wh, g)
    close(dwa, dwa_ref, 1e-4)
    close(db, g.sum((0, 1, 2)), 1e-4)
    if v is not None:
        _, dwv_ref = O.conv_same_bwd(O.pool2(v), wv, g)
        close(dwv, dwv_ref, 1e-4)


def test_wgrad_full_batch_split():
    """Full batch 128 on the biggest map with many pixel splits (atomic accumulation)."""
    import hiputil as U
    rng = np.random.default_rng(11)
    x = rng.standard_normal((128, 32, 32, 16)).astype(np.float32)
    g = rng.standard_normal((128, 32, 32, 16)).astype(np.float32)
    dwa, _, db = U.wgrad(x, g, n_split=256)
    _, ref = O.conv_same_bwd(x, np.zeros((3, 3, 16, 16)), g)
    close(dwa, ref, 1e-4)
    close(db, g.astype(np.float64).sum((0, 1, 2)), 1e-4)


@pytest.mark.parametrize('C', [16, 32, 64, 128])
def test_bn_bwd_elementwise(C):
    import hiputil as U
    rng = np.random.default_rng(C)
    n = 7
    s = rng.standard_normal((n, 4, 4, C)); dy = rng.standard_normal((n, 4, 4, C))
    gamma, beta = rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.3
    bn, cnt = U.bn_dict(s, gamma, beta)
    dz, red = U.bn_bwd_reduce(dy, s, bn, cnt)
    y, m, var = O.bn_train(s, gamma, beta)
    gx_ref, dgamma, dbeta = O.bn_train_bwd(s, gamma, m, var, np.where(y > 0, dy, 0.0))
    safe = np.abs(y) > 1e-4
    assert np.abs(dz - np.where(y > 0, dy, 0.0))[safe].max() < 1e-6
    close(red[:C], dbeta, 1e-4)
    close(red[C:], dgamma, 1e-4)
    g = U.bn_bwd_apply(dz, s, bn, cnt, red)
    close(g, gx_ref, 1e-4)


def test_pack_layout():
    """Pack kernel vs. the layout documented in include/mpnn_hip.h."""
    import hiputil as U
    rng = np.random.default_rng(5)
    for ci, co in [(3, 16), (16, 32), (32, 16), (64, 128)]:
        w = rng.standard_normal((3, 3, ci, co)).astype(np.float32)
        fw, bw = U.pack_weights([w])
        nch = (ci + 15) // 16
        ref = np.zeros((9, nch, 4, co, 4), np.float32)
        wf = w.reshape(9, ci, co)
        for c in range(ci):
            ref[:, c // 16, (c % 16) // 4, :, c % 4] = wf[:, c, :]
        assert np.array_equal(fw[0].cpu().numpy().reshape(ref.shape), ref)
        if ci % 16 == 0:
            nchb = (co + 15) // 16
            refb = np.zeros((9, nchb, 4, ci, 4), np.float32)
            for o in range(co):
                refb[:, o // 16, (o % 16) // 4, :, o % 4] = wf[::-1, :, o]
            assert np.array_equal(bw[0].cpu().numpy().reshape(refb.shape), refb)
