"""ctypes binding of libmpnn_hip.so (C ABI: include/mpnn_hip.h).

The library is the product: if it is missing or fails to load, importing a
net's execution engine raises -- there is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MPNN_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'libmpnn_hip.so')

ACT_IDENTITY, ACT_BN_BATCH, ACT_BN_MOVING, ACT_RELU = 0, 1, 2, 3
NET_SR, NET_ACTOR, NET_CRITIC = 0, 1, 2
HYP_LR, HYP_MU, HYP_TAU, HYP_EPS, HYP_KCPT, HYP_KDEC, HYP_KCRE, HYP_ARTR, HYP_N = 0, 1, 2, 3, 4, 5, 6, 7, 16
MAX_NODES, MAX_SINKS = 128, 4
BN_SLOTS = 16                  # MPNN_BN_SLOTS
SEG_INTS = 12                  # MPNN_SEG_INTS
SLAB_ITEM = 1024               # MPNN_SLAB_ITEM: elements per mpnn_slab_reduce work item
LIN_KSLICES = 8                # MPNN_LIN_KSLICES: most K-slices of one mpnn_lin_fwd record
LIN_RSPLIT = 8                 # MPNN_LIN_RSPLIT: most row groups of one mpnn_lin_bwd_rs feature block
LIN_RS_TILE = 2080             # MPNN_LIN_RS_TILE: floats of one row group's partial tile


def slab_item_size(n_split):
    """Elements per mpnn_slab_reduce item for a tensor summed over n_split slabs: the largest for which
    the kernel's slab groups (256 threads / (item / 4) quads, at most 16) leave every thread <= 16 slabs,
    i.e. one batch of loads: 1024 up to a split of 16, 64 at 256 and beyond."""
    size = SLAB_ITEM
    while size > 64 and (256 // (size // 4)) * 16 < n_split:
        size //= 2
    return size

P = C.c_void_p


class Act(C.Structure):
    _fields_ = [('x', P), ('sum', P), ('gamma', P), ('beta', P), ('m_avg', P), ('v_avg', P),
                ('eps', C.c_float), ('cnt', C.c_int), ('C', C.c_int), ('shift', C.c_int), ('mode', C.c_int),
                ('nslot', C.c_int)]


class ConvFwdArgs(C.Structure):
    _fields_ = [('a', Act), ('v', P), ('Cv', C.c_int), ('wa_pack', P), ('wv_pack', P), ('bias', P),
                ('out', P), ('pool_out', P), ('out_sum', P), ('out_nslot', C.c_int), ('n', C.c_int), ('H', C.c_int), ('W', C.c_int),
                ('Cout', C.c_int), ('idx', P), ('cnt', P)]


class BnCtx(C.Structure):
    _fields_ = [('s', P), ('bn', Act), ('red', P), ('red_nslot', C.c_int)]


class DgradHorzArgs(C.Structure):
    _fields_ = [('g', P), ('Cg', C.c_int), ('g_ctx', C.POINTER(BnCtx)), ('w_pack', P), ('dy_extra', P),
                ('prev', C.POINTER(BnCtx)),
                ('out', P), ('red_out', P), ('n', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cout', C.c_int),
                ('accumulate', C.c_int)]


class DgradVertArgs(C.Structure):
    _fields_ = [('g', P), ('Cg', C.c_int), ('g_ctx', C.POINTER(BnCtx)), ('w_pack', P), ('fine', C.POINTER(BnCtx)),
                ('fine_has_dz', C.c_int),
                ('dz_g_fine', P), ('n', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cout', C.c_int)]


class WgradArgs(C.Structure):
    _fields_ = [('a', Act), ('v', P), ('Cv', C.c_int), ('g', P), ('g_ctx', C.POINTER(BnCtx)), ('dwa', P), ('dwv', P),
                ('db', P),
                ('split_stride', C.c_long), ('n', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cout', C.c_int), ('n_split', C.c_int)]


BWD_LEVEL_MAX = 4              # MPNN_BWD_LEVEL_MAX: members of one mpnn_msconv_bwd_level launch


class BwdMember(C.Structure):
    _fields_ = [('horz', C.POINTER(DgradHorzArgs)), ('vert', C.POINTER(DgradVertArgs)), ('wgrad', C.POINTER(WgradArgs)),
                ('wg_horz', C.c_int), ('wg_vert', C.c_int)]


class LinFwdArgs(C.Structure):
    _fields_ = [('a', Act), ('HW', C.c_int), ('w', P * 2), ('b', P * 2), ('y', P * 2), ('M', C.c_int * 2),
                ('k_cpt', P), ('alpha_cpt', C.c_float), ('extra_col', C.c_int * 2), ('n', C.c_int),
                ('kpart', P), ('kcnt', P)]


class LinBwdArgs(C.Structure):
    _fields_ = [('a', Act), ('HW', C.c_int), ('w', P * 2), ('dy', P * 2), ('M', C.c_int * 2),
                ('dw', P * 2), ('db', P * 2), ('dx', P), ('k_cpt', P), ('alpha_cpt', C.c_float),
                ('extra_col', C.c_int * 2), ('n', C.c_int),
                ('dz_out', P), ('red_out', P), ('red_nslot', C.c_int), ('kpart', P), ('kcnt', P)]


class ExitTailArgs(C.Structure):
    _fields_ = [('z', P), ('y', P), ('n_cls', C.c_int), ('eps_ce', C.c_float), ('c_err', P), ('d_cor', P),
                ('h1', P), ('R', C.c_int), ('n_sinks', C.c_int),
                ('g1', P), ('b1', P), ('m1', P), ('v1', P), ('w2', P), ('bias2', P),
                ('g2', P), ('b2', P), ('m2', P), ('v2', P), ('w3', P), ('bias3', P),
                ('h2', P), ('r', P), ('r_stride', C.c_int), ('bn_save', P),
                ('bn_eps', C.c_float), ('bn_decay', C.c_float), ('mode', C.c_int), ('n', C.c_int),
                ('clear_f', P), ('n_clear_f', C.c_int), ('clear_d', P), ('n_clear_d', C.c_int), ('R2', C.c_int),
                ('hyp_src', P), ('hyp_dst', P), ('bn_eps2', C.c_float), ('bn_decay2', C.c_float)]


class ExitTailBwdArgs(C.Structure):
    _fields_ = [('f', ExitTailArgs), ('w_cerr', P), ('dr', P), ('dz', P), ('dh1', P),
                ('dg1', P), ('db1', P), ('dw2', P), ('dbias2', P), ('dg2', P), ('db2', P),
                ('dw3', P), ('dbias3', P), ('dh2', P)]


class ExitEvArgs(C.Structure):
    _fields_ = [('a', Act), ('HW', C.c_int), ('w_head', P), ('b_head', P), ('n_cls', C.c_int), ('y', P),
                ('eps_ce', C.c_float), ('c_err', P), ('d_cor', P),
                ('w1', P), ('b1', P), ('R', C.c_int), ('n_sinks', C.c_int),
                ('extra_col', C.c_int), ('k_cpt', P), ('alpha_cpt', C.c_float),
                ('g1', P), ('be1', P), ('m1', P), ('v1', P), ('w2', P), ('bias2', P),
                ('g2', P), ('be2', P), ('m2', P), ('v2', P), ('w3', P), ('bias3', P),
                ('bn_eps', C.c_float), ('r', P), ('r_stride', C.c_int),
                ('idx', P), ('cnt', P), ('n', C.c_int),
                ('child_idx', P * 4), ('child_cnt', P * 4), ('R2', C.c_int), ('z', P), ('h1', P), ('bn_eps2', C.c_float)]


class ConvNhwcFwdArgs(C.Structure):
    _fields_ = [('a', Act), ('w', P), ('bias', P), ('out', P), ('n', C.c_int), ('H', C.c_int), ('W', C.c_int),
                ('Cout', C.c_int), ('supp', C.c_int)]


class ConvNhwcDgradArgs(C.Structure):
    _fields_ = [('g', P), ('Cg', C.c_int), ('w', P), ('relu_src', P), ('scratch', P), ('dx', P),
                ('n', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('supp', C.c_int)]


class ConvNhwcWgradArgs(C.Structure):
    _fields_ = [('a', Act), ('g', P), ('dw', P), ('db', P), ('split_stride', C.c_long), ('n_split', C.c_int),
                ('n', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cout', C.c_int), ('supp', C.c_int)]


class RouteArgs(C.Structure):
    _fields_ = [('net_type', C.c_int), ('n_nodes', C.c_int), ('n_leaves', C.c_int), ('n_switches', C.c_int),
                ('max_sinks', C.c_int), ('optimistic', C.c_int), ('use_cls_err', C.c_int), ('want_grad', C.c_int),
                ('nodes', P), ('sw_children', P), ('node_ops', P), ('hyp', P), ('k_cpt_vec', P), ('r', P),
                ('c_err', P), ('d_cor', P), ('p_tr', P), ('p_ev', P), ('w_cerr', P), ('dr', P),
                ('node_stat', P), ('loss', P), ('n', C.c_int), ('n_total', C.c_int), ('stat_part', P), ('stat_ticket', P)]


class FinishNet(C.Structure):
    _fields_ = [('slabs', P), ('slab_table', P), ('n_items', C.c_int), ('item_seg', P),
                ('sums', P), ('reds', P), ('state', P), ('bn_table', P), ('n_bn', C.c_int), ('bn_opt', P), ('n_img', C.c_int),
                ('sums_keep', P), ('params', P), ('accum', P), ('grads', P), ('node_stat', P), ('hyp', P), ('talr', C.c_int),
                ('inv_n', C.c_float), ('grad_scale', C.c_float), ('w_eq', P), ('packs', P), ('plain_seg', P), ('n_plain', C.c_int)]


PREFIX_MAX = 64


class EvPrefixArgs(C.Structure):
    _fields_ = [('n', C.c_int), ('count', C.c_int), ('n_front', C.c_int), ('pad_', C.c_int),
                ('parent', C.c_int * PREFIX_MAX), ('parent_sink', C.c_int * PREFIX_MAX), ('n_sinks', C.c_int * PREFIX_MAX),
                ('r_stride', C.c_int * PREFIX_MAX), ('r', P * PREFIX_MAX), ('c_err', P * PREFIX_MAX), ('d_cor', P * PREFIX_MAX),
                ('front_parent', C.c_int * PREFIX_MAX), ('front_sink', C.c_int * PREFIX_MAX),
                ('front_idx', P * PREFIX_MAX), ('front_cnt', P * PREFIX_MAX)]


class AugmentDst(C.Structure):
    _fields_ = [('draw', P), ('x_out', P), ('y_out', P)]


_SIGS = {
    'mpnn_pack_weights': [P, P, P, C.c_int, P],
    'mpnn_step_begin': [P, P, P, C.c_int, P, C.c_long, P],
    'mpnn_msconv_fwd': [C.POINTER(ConvFwdArgs), P],
    'mpnn_msconv_fwd_group': [C.POINTER(ConvFwdArgs), P, C.c_int, P],
    'mpnn_msconv_fwd_group_rep': [C.POINTER(ConvFwdArgs), P, C.c_int, C.c_int, C.c_int, P],
    'mpnn_msconv_bwd_level_prepare_rep': [C.POINTER(BwdMember), C.c_int, C.c_int, P],
    'mpnn_msconv_bwd_level_rep': [C.POINTER(BwdMember), C.c_int, C.c_int, P, P],
    'mpnn_route_multi': [C.POINTER(RouteArgs), P, C.c_int, P],
    'mpnn_backward_finish_opt_multi': [C.POINTER(FinishNet), P, C.c_int, C.c_float, P],
    'mpnn_debug_set_trace': [P],
    'mpnn_backward_finish': [P, P, P, C.c_int, P, P, P, P, C.c_int, C.c_float, C.c_int, P, P],
    'mpnn_augment_batch': [P, P, P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P],
    'mpnn_augment_batch_multi': [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P],
    'mpnn_msconv_bwd_scale_slots': [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    'mpnn_bn_relu_fwd': [C.POINTER(Act), P, C.c_long, P],
    'mpnn_bn_bwd_reduce': [P, C.POINTER(BnCtx), P, P, C.c_long, P],
    'mpnn_bn_bwd_apply': [P, C.POINTER(BnCtx), C.c_long, P],
    'mpnn_msconv_dgrad_horz': [C.POINTER(DgradHorzArgs), P],
    'mpnn_msconv_dgrad_vert': [C.POINTER(DgradVertArgs), P],
    'mpnn_msconv_dgrad_pair': [C.POINTER(DgradHorzArgs), C.POINTER(DgradVertArgs), P],
    'mpnn_msconv_wgrad': [C.POINTER(WgradArgs), P],
    'mpnn_msconv_bwd_scale': [C.POINTER(DgradHorzArgs), C.POINTER(DgradVertArgs), C.POINTER(WgradArgs), P],
    'mpnn_msconv_bwd_level_record_size': [],
    'mpnn_msconv_bwd_level_slots': [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int],
    'mpnn_msconv_bwd_level_prepare': [C.POINTER(BwdMember), C.c_int, P],
    'mpnn_msconv_bwd_level': [C.POINTER(BwdMember), C.c_int, P, P],
    'mpnn_wgrad_tiles': [C.c_int, C.c_int, C.c_int],
    'mpnn_slab_reduce': [P, P, P, C.c_int, P],
    'mpnn_lin_fwd': [P, C.c_int, C.c_int, P],
    'mpnn_lin_bwd': [P, C.c_int, C.c_int, C.c_int, P],
    'mpnn_lin_fwd_ks': [P, C.c_int, C.c_int, C.c_int, P],
    'mpnn_lin_bwd_rs': [P, C.c_int, C.c_int, C.c_int, P],
    'mpnn_exit_tail_fwd': [P, C.c_int, C.c_int, P],
    'mpnn_exit_tail_bwd': [P, C.c_int, C.c_int, P],
    'mpnn_route': [C.POINTER(RouteArgs), P],
    'mpnn_exit_ev': [P, C.c_int, C.c_int, P],
    'mpnn_exit_ev_check': [C.POINTER(ExitEvArgs)],
    'mpnn_compact_by_branch': [P, C.c_int, P, P, P],
    'mpnn_bn_finalize': [P, P, P, P, P, C.c_int, C.c_float, C.c_int, P, P],
    'mpnn_talr_momentum_step': [P, P, P, P, C.c_int, P, P, C.c_int, C.c_float, C.c_float, P, P, P],
    'mpnn_conv_nhwc_fwd': [C.POINTER(ConvNhwcFwdArgs), P],
    'mpnn_conv_nhwc_dgrad': [C.POINTER(ConvNhwcDgradArgs), P],
    'mpnn_conv_nhwc_wgrad': [C.POINTER(ConvNhwcWgradArgs), P],
    'mpnn_backward_finish_opt': [P, P, C.c_int, P, P, P, P, P, C.c_int, P, C.c_float, C.c_int, P, P, P, P, P, P, C.c_int,
                                 C.c_float, C.c_float, P, P, P, C.c_int, P],
    'mpnn_lin_fwd_gen': [P, C.c_int, C.c_int, P],
    'mpnn_lin_bwd_gen': [P, C.c_int, C.c_int, C.c_int, P],
    'mpnn_exit_tail_fwd_gen': [P, C.c_int, C.c_int, P],
    'mpnn_exit_tail_bwd_gen': [P, C.c_int, C.c_int, P],
    'mpnn_exit_ev_gen': [P, C.c_int, C.c_int, P],
    'mpnn_exit_gen_check': [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    'mpnn_maxpool_fwd': [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P],
    'mpnn_maxpool_bwd': [P, P, P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P],
    'mpnn_set_reserved_cus': [C.c_int],
    'mpnn_debug_spin': [C.c_int, C.c_int, C.c_float, P],
    'mpnn_debug_noop': [P],
    'mpnn_draw_augmentation': [P, C.c_long, C.c_int, C.c_long, P, C.c_int, P, P],
    'mpnn_draw_augmentation_mt': [P, P, C.c_long, C.c_int, C.c_long, P, C.c_int, P],
    'mpnn_ev_prefix_walk': [C.POINTER(EvPrefixArgs), P, P],
}

_LONG = {'mpnn_draw_augmentation', 'mpnn_draw_augmentation_mt'}
EXPORTS = sorted(_SIGS) + ['mpnn_version']

_lib = None


E_SHAPE, E_ARG = -1, -2          # MPNN_E_SHAPE, MPNN_E_ARG


class HipError(RuntimeError):
    pass


def load():
    """Load libmpnn_hip.so or raise: the HIP path is the only path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError('%s not found -- build it with `make -C multipath-nn_amd/csrc` '
                           '(or __graft_entry__.build()); there is no CPU fallback' % LIB_PATH)
        # ONE HIP runtime per process: PyTorch bundles its own libamdhip64 and the library's device pointers are torch
        # allocations.  With torch loaded first the library's dependency resolves to that copy (same SONAME); loaded the
        # other way round the process holds two runtimes and the first launch fails with hipErrorNoDevice.
        try:
            import torch                        # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, sig in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = sig
            fn.restype = C.c_long if name in _LONG else C.c_int
        lib.mpnn_version.restype = C.c_char_p
        _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        raise HipError('%s failed with status %d (%s)' % (
            what, code, {-1: 'unsupported shape', -2: 'bad argument'}.get(code, 'hipError_t')))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def act(x=None, C_=0, mode=ACT_IDENTITY, shift=0, bn=None, cnt=1, nslot=BN_SLOTS):
    """Build an mpnn_act.  bn = dict(sum=, gamma=, beta=, m_avg=, v_avg=, eps=) of tensors."""
    a = Act()
    a.x = ptr(x); a.C = int(C_); a.mode = int(mode); a.shift = int(shift); a.cnt = int(cnt)
    a.nslot = int(bn.get('nslot', nslot)) if bn is not None else int(nslot)
    if bn is not None:
        a.sum = ptr(bn.get('sum')); a.gamma = ptr(bn['gamma']); a.beta = ptr(bn['beta'])
        a.m_avg = ptr(bn['m_avg']); a.v_avg = ptr(bn['v_avg']); a.eps = float(bn['eps'])
    return a


def to_device_table(records, device):
    """Upload an array of ctypes Structures as a device byte tensor."""
    import torch
    if not records:
        return None
    arr = (type(records[0]) * len(records))(*records)
    raw = bytes(memoryview(arr))
    host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    return host.to(device)
