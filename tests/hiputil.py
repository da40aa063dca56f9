"""Thin numpy-in / numpy-out drivers of the C-ABI entry points, for the parity tests."""
import ctypes as C

import numpy as np
import torch

from lib import _hip

DEV = 'cuda:0'


def dev(a, dtype=torch.float32):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a)).to(DEV, dtype)


def stream():
    return torch.cuda.current_stream().cuda_stream


def pack_sizes(cin, cout):
    return 9 * ((cin + 15) // 16) * 16 * cout, 9 * ((cout + 15) // 16) * 16 * cin


def pack_weights(ws, want_bwd=True):
    """ws: list of HWIO float arrays.  Returns (list of fwd packs, list of bwd packs) as device tensors."""
    lib = _hip.load()
    flat = np.concatenate([w.reshape(-1) for w in ws]).astype(np.float32)
    desc, off, poff = [], 0, 0
    spans = []
    for w in ws:
        _, _, ci, co = w.shape
        fs, bs = pack_sizes(ci, co)
        has_b = want_bwd and ci % 16 == 0
        desc += [off, poff, poff + fs if has_b else -1, ci, co, 0]
        spans.append((poff, fs, poff + fs, bs if has_b else 0))
        off += w.size
        poff += fs + (bs if has_b else 0)
    params, packs = dev(flat), torch.zeros(poff, device=DEV)
    d = dev(np.array(desc, np.int32), torch.int32)
    _hip.check(lib.mpnn_pack_weights(params.data_ptr(), packs.data_ptr(), d.data_ptr(), len(ws), stream()), 'pack')
    torch.cuda.synchronize()
    return ([packs[a:a + n] for a, n, _, _ in spans], [packs[b:b + m] if m else None for _, _, b, m in spans])


def slots(v):
    """A statistics vector [2C] spread (unevenly, on purpose) over the [SLOTS][2C] layout."""
    v = np.asarray(v, np.float64)
    out = np.zeros((_hip.BN_SLOTS, v.size))
    w = np.random.default_rng(7).random(_hip.BN_SLOTS); w /= w.sum()
    out[:] = w[:, None] * v[None, :]
    out[0] += v - out.sum(0)
    return out


def unslot(t, c2):
    return t.cpu().numpy().reshape(_hip.BN_SLOTS, c2).sum(0)


def bn_dict(x, gamma, beta, m_avg=None, v_avg=None, eps=1e-6):
    """Device BatchNorm context for pre-BN array x (statistics over all leading dims)."""
    c = x.shape[-1]
    x64 = np.asarray(x, np.float64).reshape(-1, c)
    sums = slots(np.concatenate([x64.sum(0), (x64 ** 2).sum(0)]))
    d = dict(sum=dev(sums, torch.float64), gamma=dev(gamma), beta=dev(beta),
             m_avg=dev(np.zeros(c) if m_avg is None else m_avg),
             v_avg=dev(np.ones(c) if v_avg is None else v_avg), eps=eps)
    return d, x64.shape[0]


def pool2_np(v):
    n, h, w, c = v.shape
    return np.ascontiguousarray(v.reshape(n, h // 2, 2, w // 2, 2, c).max(axis=(2, 4)))


def conv_fwd(x, wh, b, v=None, wv=None, bn=None, mode=_hip.ACT_IDENTITY, shift=0, bn_cnt=1, want_pool=False, group=False):
    """x: [n, H<<shift, W<<shift, Ca]; v: the UNPOOLED finer map [n, 2H, 2W, Cv] (pooled here, as its
    producer would).  Returns (out, out_sum[, pooled out])."""
    if v is not None:
        v = pool2_np(np.asarray(v, np.float32))
    lib = _hip.load()
    n = x.shape[0]
    H, W = x.shape[1] >> shift, x.shape[2] >> shift
    co = wh.shape[3]
    fw, _ = pack_weights([wh] + ([wv] if wv is not None else []), want_bwd=False)
    xd, vd, bd = dev(x), dev(v), dev(b)
    out = torch.empty((n, H, W, co), device=DEV)
    osum = torch.zeros(_hip.BN_SLOTS * 2 * co, device=DEV, dtype=torch.float64)
    a = _hip.ConvFwdArgs()
    a.a = _hip.act(xd, x.shape[3], mode, shift, bn, bn_cnt)
    a.v = _hip.ptr(vd); a.Cv = v.shape[3] if v is not None else 0
    a.wa_pack = fw[0].data_ptr(); a.wv_pack = fw[1].data_ptr() if wv is not None else None
    a.bias = bd.data_ptr(); a.out = out.data_ptr(); a.out_sum = osum.data_ptr(); a.out_nslot = _hip.BN_SLOTS
    pool = torch.full((n, H // 2, W // 2, co), 9.0, device=DEV) if want_pool else None
    a.pool_out = _hip.ptr(pool)
    a.n, a.H, a.W, a.Cout = n, H, W, co
    if group:            # the same conv as a one-member wavefront group (device table; K-split body on deep small maps)
        arr = (_hip.ConvFwdArgs * 1)(a)
        tab = _hip.to_device_table([a], DEV)
        _hip.check(lib.mpnn_msconv_fwd_group(arr, tab.data_ptr(), 1, stream()), 'msconv_fwd_group')
    else:
        _hip.check(lib.mpnn_msconv_fwd(C.byref(a), stream()), 'msconv_fwd')
    torch.cuda.synchronize()
    if want_pool:
        return out.cpu().numpy(), unslot(osum, 2 * co), pool.cpu().numpy()
    return out.cpu().numpy(), unslot(osum, 2 * co)


def bn_ctx(s_dev, C_, bn, cnt, mode=_hip.ACT_BN_BATCH, red=None):
    ctx = _hip.BnCtx()
    ctx.s = s_dev.data_ptr()
    ctx.bn = _hip.act(None, C_, mode, 0, bn, cnt)
    ctx.red = _hip.ptr(red)
    ctx.red_nslot = _hip.BN_SLOTS
    return ctx


def dgrad_horz(g, w, s_prev=None, bn=None, cnt=1, extra=None):
    """g: [n,H,W,Cout_fwd]; w: HWIO [3,3,Cin,Cout_fwd].  Returns (out, red) ; red None if raw."""
    lib = _hip.load()
    n, H, W, cg = g.shape
    ci = w.shape[2]
    _, bw = pack_weights([w])
    gd, ed = dev(g), dev(extra)
    out = torch.empty((n, H, W, ci), device=DEV)
    a = _hip.DgradHorzArgs()
    a.g = gd.data_ptr(); a.Cg = cg; a.w_pack = bw[0].data_ptr(); a.dy_extra = _hip.ptr(ed)
    a.out = out.data_ptr(); a.n, a.H, a.W, a.Cout = n, H, W, ci
    red = None
    keep = []
    if s_prev is not None:
        sd = dev(s_prev)
        red = torch.zeros(_hip.BN_SLOTS * 2 * ci, device=DEV, dtype=torch.float64)
        ctx = bn_ctx(sd, ci, bn, cnt)
        keep += [sd, ctx]
        a.prev = C.pointer(ctx); a.red_out = red.data_ptr()
    _hip.check(lib.mpnn_msconv_dgrad_horz(C.byref(a), stream()), 'dgrad_horz')
    torch.cuda.synchronize()
    return out.cpu().numpy(), (None if red is None else unslot(red, 2 * ci))


def dgrad_vert(g, w, s_fine, bn, cnt, dz_fine=None, red=None):
    """g: coarse grad [n,H,W,Cg]; w: HWIO [3,3,Cf,Cg]; s_fine/dz_fine: [n,2H,2W,Cf]."""
    lib = _hip.load()
    n, H, W, cg = g.shape
    cf = w.shape[2]
    _, bw = pack_weights([w])
    gd, sd = dev(g), dev(s_fine)
    buf = dev(dz_fine) if dz_fine is not None else torch.full((n, 2 * H, 2 * W, cf), 7.0, device=DEV)
    redd = dev(slots(red), torch.float64) if red is not None else None
    ctx = bn_ctx(sd, cf, bn, cnt, red=redd)
    a = _hip.DgradVertArgs()
    a.g = gd.data_ptr(); a.Cg = cg; a.w_pack = bw[0].data_ptr(); a.fine = C.pointer(ctx)
    a.fine_has_dz = 1 if dz_fine is not None else 0
    a.dz_g_fine = buf.data_ptr(); a.n, a.H, a.W, a.Cout = n, H, W, cf
    _hip.check(lib.mpnn_msconv_dgrad_vert(C.byref(a), stream()), 'dgrad_vert')
    torch.cuda.synchronize()
    return buf.cpu().numpy()


def wgrad(x, g, v=None, bn=None, mode=_hip.ACT_IDENTITY, shift=0, bn_cnt=1, n_split=7):
    """Partial sums into a slab + mpnn_slab_reduce (n_split > 1) or straight into the gradients."""
    lib = _hip.load()
    if v is not None:
        v = pool2_np(np.asarray(v, np.float32))          # the producer's pooled map
    n = x.shape[0]
    H, W = x.shape[1] >> shift, x.shape[2] >> shift
    ca, co = x.shape[3], g.shape[3]
    cv = v.shape[3] if v is not None else 0
    xd, gd, vd = dev(x), dev(g), dev(v)
    sizes = [9 * ca * co, 9 * cv * co, co]
    offs = [0, sizes[0], sizes[0] + sizes[1]]
    total = sum(sizes)
    stride = (total + 3) // 4 * 4
    n_split = max(1, min(n_split, lib.mpnn_wgrad_tiles(n, H, W)))
    grads = torch.full((total,), 3.0, device=DEV)          # poison: every element must be written
    slab = grads if n_split == 1 else torch.full((n_split * stride,), 5.0, device=DEV)
    a = _hip.WgradArgs()
    a.a = _hip.act(xd, ca, mode, shift, bn, bn_cnt)
    a.v = _hip.ptr(vd); a.Cv = cv
    a.g = gd.data_ptr()
    a.dwa = slab[offs[0]:].data_ptr(); a.dwv = slab[offs[1]:].data_ptr() if cv else None
    a.db = slab[offs[2]:].data_ptr()
    a.split_stride = stride if n_split > 1 else 0
    a.n, a.H, a.W, a.Cout, a.n_split = n, H, W, co, n_split
    _hip.check(lib.mpnn_msconv_wgrad(C.byref(a), stream()), 'wgrad')
    if n_split > 1:
        tab = []
        for o, sz in zip(offs, sizes):
            item = _hip.slab_item_size(n_split)
            for k in range(0, sz, item):
                tab += [o + k, o + k, min(item, sz - k), n_split, stride, 0]
        t = dev(np.array(tab, np.int32), torch.int32)
        _hip.check(lib.mpnn_slab_reduce(slab.data_ptr(), grads.data_ptr(), t.data_ptr(), len(tab) // 6, stream()),
                   'slab_reduce')
    torch.cuda.synchronize()
    out = grads.cpu().numpy()
    dwa = out[:sizes[0]].reshape(3, 3, ca, co)
    dwv = out[offs[1]:offs[2]].reshape(3, 3, cv, co) if cv else None
    return dwa, dwv, out[offs[2]:]


def bn_bwd_reduce(dy, s, bn, cnt):
    lib = _hip.load()
    c = s.shape[-1]
    dyd, sd = dev(dy), dev(s)
    dz = torch.empty_like(dyd)
    red = torch.zeros(_hip.BN_SLOTS * 2 * c, device=DEV, dtype=torch.float64)
    ctx = bn_ctx(sd, c, bn, cnt)
    _hip.check(lib.mpnn_bn_bwd_reduce(dyd.data_ptr(), C.byref(ctx), dz.data_ptr(), red.data_ptr(),
                                      dy.size // c, stream()), 'bn_bwd_reduce')
    torch.cuda.synchronize()
    return dz.cpu().numpy(), unslot(red, 2 * c)


def bn_bwd_apply(dz, s, bn, cnt, red):
    lib = _hip.load()
    c = s.shape[-1]
    dzd, sd, redd = dev(dz), dev(s), dev(slots(red), torch.float64)
    ctx = bn_ctx(sd, c, bn, cnt, red=redd)
    _hip.check(lib.mpnn_bn_bwd_apply(dzd.data_ptr(), C.byref(ctx), dz.size // c, stream()), 'bn_bwd_apply')
    torch.cuda.synchronize()
    return dzd.cpu().numpy()
