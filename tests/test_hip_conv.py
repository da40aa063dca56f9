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


def test_step_begin_packs_and_clears():
    """mpnn_step_begin = mpnn_pack_weights + clearing the accumulator arena, in one launch."""
    import ctypes as C
    import torch
    import hiputil as U
    from lib import _hip
    lib = _hip.load()
    rng = np.random.default_rng(11)
    ws = [rng.standard_normal((3, 3, ci, co)).astype(np.float32) for ci, co in [(3, 16), (16, 32), (32, 64)]]
    fw, bw = U.pack_weights(ws)
    flat = np.concatenate([w.reshape(-1) for w in ws]).astype(np.float32)
    desc, off, poff = [], 0, 0
    for w in ws:
        _, _, ci, co = w.shape
        fs, bs = U.pack_sizes(ci, co)
        has_b = ci % 16 == 0
        desc += [off, poff, poff + fs if has_b else -1, ci, co, 0]
        off += w.size; poff += fs + (bs if has_b else 0)
    params, packs = U.dev(flat), torch.zeros(poff, device=U.DEV)
    d = U.dev(np.array(desc, np.int32), torch.int32)
    arena = torch.full((4096 + 8,), 7.0, device=U.DEV)
    z = arena[4:4 + 4096]                                   # 16-byte aligned interior slice
    _hip.check(lib.mpnn_step_begin(params.data_ptr(), packs.data_ptr(), d.data_ptr(), len(ws),
                                   z.data_ptr(), z.numel() * 4, U.stream()), 'step_begin')
    torch.cuda.synchronize()
    ref = torch.cat([t for pair in zip(fw, bw) for t in pair if t is not None])
    assert torch.equal(packs, ref)
    assert float(z.abs().max()) == 0.0 and float(arena[:4].min()) == 7.0 and float(arena[4 + 4096:].min()) == 7.0
    # misaligned / odd sizes are refused
    assert lib.mpnn_step_begin(params.data_ptr(), packs.data_ptr(), d.data_ptr(), len(ws),
                               arena[1:].data_ptr(), 64, U.stream()) == _hip.E_ARG


@pytest.mark.parametrize('split', [1, 3, 16, 17, 40, 256, 515])
def test_slab_reduce_items_and_groups(split):
    """mpnn_slab_reduce against a float64 sum for every item size / slab-group count the planner can
    produce (`_hip.slab_item_size`), ragged last items, a tensor whose offset is not 16-byte aligned
    (scalar path) and a deliberately oversized item (more than 16 slabs per thread: the kernel's loop)."""
    import torch
    import hiputil as U
    from lib import _hip
    lib = _hip.load()
    rng = np.random.default_rng(split)
    sizes = [1024 * 3 + 20, 433, 7, 64]
    stride = (sum(sizes) + 5 + 3) // 4 * 4
    slab = torch.tensor(rng.standard_normal(split * stride).astype(np.float32), device=U.DEV)
    grads = torch.full((stride,), 7.0, device=U.DEV)
    tab, off, want = [], 0, []
    for ti, sz in enumerate(sizes):
        if ti == 2:
            off += 1                                   # misaligned tensor: the scalar path
        item = _hip.slab_item_size(split) if ti != 3 else 64
        for k in range(0, sz, item):
            tab += [off + k, off + k, min(item, sz - k), split, stride, 0]
        want.append((off, sz))
        off += sz
    tab += [0, want[0][0], 1024, split, stride, 0]     # item 0 again as a full-size item whatever the split
    t = torch.tensor(tab, dtype=torch.int32, device=U.DEV)
    _hip.check(lib.mpnn_slab_reduce(slab.data_ptr(), grads.data_ptr(), t.data_ptr(), len(tab) // 6, U.stream()), 'slab_reduce')
    torch.cuda.synchronize()
    ref = slab.cpu().numpy().astype(np.float64).reshape(split, stride).sum(0)
    got = grads.cpu().numpy()
    for o, sz in want:
        assert np.abs(got[o:o + sz] - ref[o:o + sz]).max() <= 2e-6 * np.sqrt(split) * (1 + np.abs(ref[o:o + sz]).max()), (split, o, sz)
    untouched = np.ones(stride, bool)
    for o, sz in want:
        untouched[o:o + sz] = False
    assert (got[untouched] == 7.0).all()


def test_bwd_scale_slots_query():
    from lib import _hip
    import torch
    torch.zeros(1, device='cuda')
    lib = _hip.load()
    for H, Cc in ((4, 128), (8, 64), (16, 32), (32, 16)):
        s = lib.mpnn_msconv_bwd_scale_slots(H, H, Cc, 1, 1, 4096)
        assert s >= 256 and s % 256 == 0, (H, Cc, s)          # whole workgroups per CU x 256 CUs
    # a 64-channel layer whose input gradients cannot fill one workgroup per CU: two workgroups per CU
    assert lib.mpnn_msconv_bwd_scale_slots(4, 4, 128, 1, 0, 256) == 512
    assert lib.mpnn_msconv_bwd_scale_slots(8, 8, 64, 1, 0, 512) >= 512
    assert lib.mpnn_msconv_bwd_scale_slots(5, 5, 16, 1, 0, 0) == _hip.E_SHAPE
    # level launches: the variant covering a set of member shapes, and its record size
    import ctypes as C
    arr = lambda *v: (C.c_int * len(v))(*v)
    s = lib.mpnn_msconv_bwd_level_slots(arr(32, 8), arr(32, 8), arr(16, 16), 2)
    assert s >= 512 and s % 256 == 0
    assert lib.mpnn_msconv_bwd_level_slots(arr(5), arr(5), arr(16), 1) == _hip.E_SHAPE
    assert lib.mpnn_msconv_bwd_level_record_size() > 0


@pytest.mark.parametrize('C_', [32, 128])
def test_lin_bwd_fused_bn_reduce(C_):
    """lin_bwd with dz_out/red_out set = lin_bwd followed by mpnn_bn_bwd_reduce on its dX."""
    import ctypes as C
    import torch
    import hiputil as U
    from lib import _hip
    lib = _hip.load()
    rng = np.random.default_rng(C_)
    n, HW, M = 37, 16, 10
    K = HW * C_
    s = rng.standard_normal((n, 4, 4, C_)).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, C_), rng.standard_normal(C_) * 0.3
    bn, cnt = U.bn_dict(s, gamma, beta)
    w = (rng.standard_normal((K, M)) / np.sqrt(K)).astype(np.float32)
    dy = rng.standard_normal((n, M)).astype(np.float32)
    sd, wd, dyd = U.dev(s), U.dev(w), U.dev(dy)

    def run(fused):
        a = _hip.LinBwdArgs()
        a.a = U.bn_ctx(sd, C_, bn, cnt).bn
        a.a.x = sd.data_ptr()
        a.HW, a.n = HW, n
        a.w[0], a.dy[0], a.M[0] = wd.data_ptr(), dyd.data_ptr(), M
        dw, db = torch.zeros(K * M, device=U.DEV), torch.zeros(M, device=U.DEV)
        a.dw[0], a.db[0] = dw.data_ptr(), db.data_ptr()
        dx = torch.zeros(n * K, device=U.DEV)
        dz = torch.zeros(n * K, device=U.DEV)
        red = torch.zeros(_hip.BN_SLOTS * 2 * C_, device=U.DEV, dtype=torch.float64)
        a.dx = dx.data_ptr()
        if fused:
            a.dz_out, a.red_out, a.red_nslot = dz.data_ptr(), red.data_ptr(), _hip.BN_SLOTS
        tab = _hip.to_device_table([a], U.DEV)
        _hip.check(lib.mpnn_lin_bwd(tab.data_ptr(), 1, n, K, U.stream()), 'lin_bwd')
        torch.cuda.synchronize()
        return dx.cpu().numpy(), dz.cpu().numpy(), U.unslot(red, 2 * C_), dw.cpu().numpy()

    dx0, _, _, dw0 = run(False)
    dx1, dz1, red1, dw1 = run(True)
    assert np.array_equal(dx0, dx1)
    close(dw1, dw0, 1e-5)
    dz_ref, red_ref = U.bn_bwd_reduce(dx0.reshape(n, 4, 4, C_), s, bn, cnt)
    y, _, _ = O.bn_train(s.astype(np.float64), gamma, beta)
    safe = (np.abs(y) > 1e-4).reshape(-1)
    assert np.abs(dz1 - dz_ref.reshape(-1))[safe].max() < 1e-6
    close(red1, red_ref, 1e-4)


@pytest.mark.parametrize('shape', [(4, 64, 64, 64), (4, 64, 0, 128), (4, 128, 0, 128), (8, 64, 0, 64), (8, 32, 32, 64), (16, 32, 0, 32),
                                   (8, 64, 64, 64), (8, 128, 0, 32), (4, 128, 64, 64), (4, 64, 128, 16)])
@pytest.mark.parametrize('ks4', ['0', '1'])
def test_one_member_group_equals_single_launch(shape, ks4, monkeypatch):
    """mpnn_msconv_fwd_group with ONE member (the deep 4x4 / 8x8 shapes take the K-split body: 512
    threads, the two halves' partial sums meet in LDS; ks4: the opt-in four-way split, 1 024 threads, for inputs of
    at least 128 channels) against mpnn_msconv_fwd and the oracle, on a ragged batch, with BatchNorm on load and the
    pooled output."""
    import hiputil as U
    monkeypatch.setenv('MPNN_FWD_KSPLIT4', ks4)
    H, ca, cv, co = shape
    rng = np.random.default_rng(sum(shape))
    n = 11
    x = rng.standard_normal((n, H, H, ca)).astype(np.float32)
    wh = (rng.standard_normal((3, 3, ca, co)) / np.sqrt(9 * ca)).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    v = rng.standard_normal((n, 2 * H, 2 * H, cv)).astype(np.float32) if cv else None
    wv = (rng.standard_normal((3, 3, cv, co)) / np.sqrt(9 * cv)).astype(np.float32) if cv else None
    gamma, beta = rng.uniform(0.5, 1.5, ca), rng.standard_normal(ca) * 0.3
    bn, cnt = U.bn_dict(x, gamma, beta)
    want_pool = H >= 8
    one = U.conv_fwd(x, wh, b, v, wv, bn, _hip_mode(), 0, cnt, want_pool=want_pool)
    grp = U.conv_fwd(x, wh, b, v, wv, bn, _hip_mode(), 0, cnt, want_pool=want_pool, group=True)
    scale = 1 + np.abs(one[0]).max()
    assert np.abs(grp[0] - one[0]).max() <= 2e-5 * scale           # (different summation order over K)
    close(grp[1], one[1], 1e-4)
    if want_pool:
        assert np.abs(grp[2] - one[2]).max() <= 2e-5 * scale
    y, _, _ = O.bn_train(x.astype(np.float64), gamma, beta)
    ref = O.conv_same(np.maximum(y, 0), wh.astype(np.float64)) + b
    if cv:
        ref = ref + O.conv_same(U.pool2_np(v).astype(np.float64), wv.astype(np.float64))
    close(grp[0], ref, 2e-5)


@pytest.mark.parametrize('case', [(32, 32, 3, 0, 11), (32, 32, 1, 0, 5), (16, 16, 4, 0, 7), (32, 32, 3, 1, 3), (16, 48, 2, 0, 4), (32, 32, 3, 0, 64)])
@pytest.mark.parametrize('want_pool', [False, True])
def test_first_conv_kernel(case, want_pool):
    """The first conv of a net (image -> 16 channels, no operand V) runs in its own wave-per-tile kernel when
    it is launched as a one-member group (conv_first.hip): outputs and the pooled map BIT-IDENTICAL to the
    general body's (mpnn_msconv_fwd), the statistics the same sums in another order; against the oracle."""
    import hiputil as U
    from lib import _hip
    H, W, ca, shift, n = case
    rng = np.random.default_rng(H * 7 + W + ca + shift + n)
    x = rng.standard_normal((n, H << shift, W << shift, ca)).astype(np.float32)
    wh = (rng.standard_normal((3, 3, ca, 16)) / np.sqrt(9 * ca)).astype(np.float32)
    b = (rng.standard_normal(16) * 0.1).astype(np.float32)
    one = U.conv_fwd(x, wh, b, None, None, None, _hip.ACT_IDENTITY, shift, 1, want_pool=want_pool)
    grp = U.conv_fwd(x, wh, b, None, None, None, _hip.ACT_IDENTITY, shift, 1, want_pool=want_pool, group=True)
    assert np.array_equal(grp[0], one[0])
    if want_pool:
        assert np.array_equal(grp[2], one[2])
    close(grp[1], one[1], 1e-5)
    ref = O.conv_same(x[:, ::1 << shift, ::1 << shift, :].astype(np.float64), wh.astype(np.float64)) + b
    close(grp[0], ref, 2e-5)
    close(grp[1][:16], ref.sum(axis=(0, 1, 2)), 1e-4)
    close(grp[1][16:], (ref * ref).sum(axis=(0, 1, 2)), 1e-4)


@pytest.mark.parametrize('case', [(32, 32, 512, 16), (16, 16, 515, 32), (16, 48, 512, 16), (8, 16, 640, 64), (12, 16, 512, 16), (20, 32, 513, 16)])
@pytest.mark.parametrize('bn', [False, True])
@pytest.mark.parametrize('want_pool', [False, True])
def test_strip_conv_kernel(case, bn, want_pool):
    """16 -> 16 k channels on a big map without operand V runs in the wave-per-strip body when it is a member of a
    group launch of an evaluation-size batch (>= 512 samples; conv_strip.h): outputs and the pooled map BIT-IDENTICAL to the general body's (mpnn_msconv_fwd),
    the statistics the same sums in another order; against the oracle."""
    import hiputil as U
    from lib import _hip
    H, W, n, co = case
    rng = np.random.default_rng(H * 7 + W + n)
    x = rng.standard_normal((n, H, W, 16)).astype(np.float32)
    wh = (rng.standard_normal((3, 3, 16, co)) / 12).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, 16), rng.standard_normal(16) * 0.3
    bnd, cnt = U.bn_dict(x, gamma, beta) if bn else (None, 1)
    mode = _hip.ACT_BN_BATCH if bn else _hip.ACT_IDENTITY
    one = U.conv_fwd(x, wh, b, None, None, bnd, mode, 0, cnt, want_pool=want_pool)
    grp = U.conv_fwd(x, wh, b, None, None, bnd, mode, 0, cnt, want_pool=want_pool, group=True)
    assert np.array_equal(grp[0], one[0])
    if want_pool:
        assert np.array_equal(grp[2], one[2])
    close(grp[1], one[1], 1e-5)
    xs = x.astype(np.float64)
    if bn:
        y, _, _ = O.bn_train(xs, gamma, beta)
        xs = np.maximum(y, 0)
    ref = O.conv_same(xs, wh.astype(np.float64)) + b
    close(grp[0], ref, 2e-5)


@pytest.mark.parametrize('case', [(16, 16, 512, 16, 16, 16), (16, 16, 512, 32, 0, 32), (16, 32, 513, 16, 16, 32), (8, 16, 640, 32, 16, 16), (12, 16, 512, 32, 0, 16)])
@pytest.mark.parametrize('bn', [False, True])
def test_strip_conv_kernel_two_or_three_chunks(case, bn):
    """32-channel inputs and / or the pooled finer map V (2-3 sixteen-channel chunks) on a big map at an evaluation-size
    batch: the multi-chunk strip body (weights in LDS; conv_strip.h).  Its summation order differs from the general
    body's (dy, chunk, ... against chunk, dy, ...): equal to fp32 rounding, and against the oracle; the pooled map is
    the exact max-pool of its own output."""
    import hiputil as U
    from lib import _hip
    H, W, n, ca, cv, co = case
    rng = np.random.default_rng(H * 7 + W + n + ca + cv)
    x = rng.standard_normal((n, H, W, ca)).astype(np.float32)
    v = rng.standard_normal((n, 2 * H, 2 * W, cv)).astype(np.float32) if cv else None
    wh = (rng.standard_normal((3, 3, ca, co)) / np.sqrt(9 * ca)).astype(np.float32)
    wv = (rng.standard_normal((3, 3, cv, co)) / np.sqrt(9 * cv)).astype(np.float32) if cv else None
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, ca), rng.standard_normal(ca) * 0.3
    bnd, cnt = U.bn_dict(x, gamma, beta) if bn else (None, 1)
    mode = _hip.ACT_BN_BATCH if bn else _hip.ACT_IDENTITY
    one = U.conv_fwd(x, wh, b, v, wv, bnd, mode, 0, cnt, want_pool=True)
    grp = U.conv_fwd(x, wh, b, v, wv, bnd, mode, 0, cnt, want_pool=True, group=True)
    scale = 1 + np.abs(one[0]).max()
    assert np.abs(grp[0] - one[0]).max() <= 2e-5 * scale
    assert np.array_equal(grp[2], U.pool2_np(grp[0]))
    close(grp[1], one[1], 1e-4)
    xs = x.astype(np.float64)
    if bn:
        y, _, _ = O.bn_train(xs, gamma, beta)
        xs = np.maximum(y, 0)
    ref = O.conv_same(xs, wh.astype(np.float64)) + b
    if cv:
        ref = ref + O.conv_same(U.pool2_np(v).astype(np.float64), wv.astype(np.float64))
    close(grp[0], ref, 2e-5)


@pytest.mark.parametrize('case', [(16, 16, 512, 3, 16, 16, 1), (16, 32, 513, 1, 16, 32, 1), (16, 16, 512, 3, 32, 16, 2)])
def test_strip_conv_kernel_image_plus_v(case):
    """Block 0's coarser scales on a big map: the 1..3-channel pyramid image (strided pick, shift) plus the pooled finer
    map V, at an evaluation-size batch -- the image form of the multi-chunk strip body (conv_strip.h, SMA): against
    the general body to fp32 summation order and against the oracle; the pooled map is the max-pool of its own output."""
    import hiputil as U
    from lib import _hip
    H, W, n, ca, cv, co, shift = case
    rng = np.random.default_rng(H * 7 + W + n + ca + cv + shift)
    x = rng.standard_normal((n, H << shift, W << shift, ca)).astype(np.float32)
    v = rng.standard_normal((n, 2 * H, 2 * W, cv)).astype(np.float32)
    wh = (rng.standard_normal((3, 3, ca, co)) / np.sqrt(9 * ca)).astype(np.float32)
    wv = (rng.standard_normal((3, 3, cv, co)) / np.sqrt(9 * cv)).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    one = U.conv_fwd(x, wh, b, v, wv, None, _hip.ACT_IDENTITY, shift, 1, want_pool=True)
    grp = U.conv_fwd(x, wh, b, v, wv, None, _hip.ACT_IDENTITY, shift, 1, want_pool=True, group=True)
    scale = 1 + np.abs(one[0]).max()
    assert np.abs(grp[0] - one[0]).max() <= 2e-5 * scale
    assert np.array_equal(grp[2], U.pool2_np(grp[0]))
    close(grp[1], one[1], 1e-4)
    xs = x[:, ::1 << shift, ::1 << shift, :].astype(np.float64)
    ref = O.conv_same(xs, wh.astype(np.float64)) + b + O.conv_same(U.pool2_np(v).astype(np.float64), wv.astype(np.float64))
    close(grp[0], ref, 2e-5)


@pytest.mark.parametrize('shape', [(8, 32, 32, 64), (4, 64, 64, 64), (4, 64, 0, 128), (8, 32, 0, 32)])
def test_wide_group_equals_single_launch(shape):
    """Evaluation batches (capacity >= 1024, moving-average BatchNorm): a group whose members are 8x8 / 4x4 convs with
    Cout % 32 == 0 runs 32-channel output tiles (fwd_group_k<.., WIDE>).  Same contraction order per output element:
    bit-identical to mpnn_msconv_fwd, pooled map included; against the oracle on the first samples."""
    import hiputil as U
    from lib import _hip
    H, ca, cv, co = shape
    rng = np.random.default_rng(sum(shape) + 1)
    n = 1027
    x = rng.standard_normal((n, H, H, ca)).astype(np.float32)
    wh = (rng.standard_normal((3, 3, ca, co)) / np.sqrt(9 * ca)).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    v = rng.standard_normal((n, 2 * H, 2 * H, cv)).astype(np.float32) if cv else None
    wv = (rng.standard_normal((3, 3, cv, co)) / np.sqrt(9 * cv)).astype(np.float32) if cv else None
    gamma, beta = rng.uniform(0.5, 1.5, ca), rng.standard_normal(ca) * 0.3
    m_avg, v_avg = rng.standard_normal(ca) * 0.2, rng.uniform(0.5, 1.5, ca)
    bn, cnt = U.bn_dict(x, gamma, beta, m_avg, v_avg)
    want_pool = H >= 8
    one = U.conv_fwd(x, wh, b, v, wv, bn, _hip.ACT_BN_MOVING, 0, cnt, want_pool=want_pool)
    grp = U.conv_fwd(x, wh, b, v, wv, bn, _hip.ACT_BN_MOVING, 0, cnt, want_pool=want_pool, group=True)
    assert np.array_equal(grp[0], one[0])
    if want_pool:
        assert np.array_equal(grp[2], one[2])
    k = 16
    y = gamma * (x[:k].astype(np.float64) - m_avg) / np.sqrt(v_avg + 1e-6) + beta
    ref = O.conv_same(np.maximum(y, 0), wh.astype(np.float64)) + b
    if cv:
        ref = ref + O.conv_same(U.pool2_np(v[:k]).astype(np.float64), wv.astype(np.float64))
    close(grp[0][:k], ref, 2e-5)


def _hip_mode():
    from lib import _hip
    return _hip.ACT_BN_BATCH
