"""CPU: host-side logic and the C-ABI boundary (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'mpnn_hip.h')


def test_library_exports_every_declared_symbol():
    from lib import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    declared = set(re.findall(r'^(?:int|long|const char \*)\s*(mpnn_\w+)\(', open(HEADER).read(), re.M))
    assert declared and declared == set(_hip.EXPORTS), declared ^ set(_hip.EXPORTS)
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert getattr(lib, name) is not None
    lib.mpnn_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.mpnn_version()
    # a pure-host entry point may be called without a GPU
    assert lib.mpnn_wgrad_tiles(128, 32, 32) == 2048 and lib.mpnn_wgrad_tiles(128, 4, 4) == 32
    assert lib.mpnn_wgrad_tiles(5, 4, 4) == 2 and lib.mpnn_wgrad_tiles(1, 7, 7) == -1


def test_ctypes_structs_match_the_c_layout():
    """sizeof/offsetof of every argument record, as gcc sees the header, vs. the ctypes mirror."""
    from lib import _hip
    pairs = [('mpnn_act', _hip.Act), ('mpnn_conv_fwd_args', _hip.ConvFwdArgs), ('mpnn_bn_ctx', _hip.BnCtx),
             ('mpnn_dgrad_horz_args', _hip.DgradHorzArgs), ('mpnn_dgrad_vert_args', _hip.DgradVertArgs),
             ('mpnn_wgrad_args', _hip.WgradArgs), ('mpnn_bwd_member', _hip.BwdMember), ('mpnn_lin_fwd_args', _hip.LinFwdArgs),
             ('mpnn_lin_bwd_args', _hip.LinBwdArgs), ('mpnn_exit_tail_args', _hip.ExitTailArgs),
             ('mpnn_exit_tail_bwd_args', _hip.ExitTailBwdArgs), ('mpnn_route_args', _hip.RouteArgs),
             ('mpnn_exit_ev_args', _hip.ExitEvArgs), ('mpnn_conv_nhwc_fwd_args', _hip.ConvNhwcFwdArgs),
             ('mpnn_conv_nhwc_dgrad_args', _hip.ConvNhwcDgradArgs), ('mpnn_conv_nhwc_wgrad_args', _hip.ConvNhwcWgradArgs),
             ('mpnn_finish_net', _hip.FinishNet), ('mpnn_ev_prefix_args', _hip.EvPrefixArgs)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mpnn_hip.h"', 'int main(void){']
    for cname, cls in pairs:
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, f))
        lines.append('printf("\\n");')
    lines.append('return 0;}')
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, 'a.c'), os.path.join(d, 'a.out')
        open(src, 'w').write('\n'.join(lines))
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), src, '-o', exe])
        out = subprocess.check_output([exe]).decode().splitlines()
    for (cname, cls), line in zip(pairs, out):
        nums = [int(x) for x in line.split()[1:]]
        assert nums[0] == ctypes.sizeof(cls), (cname, nums[0], ctypes.sizeof(cls))
        assert nums[1:] == [getattr(cls, f).offset for f, _ in cls._fields_], cname
    for name, val in re.findall(r'#define (MPNN_\w+)\s+\(?(-?\d+)\)?', open(HEADER).read()):
        py = {'MPNN_BN_SLOTS': _hip.BN_SLOTS, 'MPNN_MAX_NODES': _hip.MAX_NODES, 'MPNN_MAX_SINKS': _hip.MAX_SINKS,
              'MPNN_HYP_N': _hip.HYP_N, 'MPNN_BWD_LEVEL_MAX': _hip.BWD_LEVEL_MAX, 'MPNN_HYP_TAU': _hip.HYP_TAU, 'MPNN_HYP_EPS': _hip.HYP_EPS,
              'MPNN_NET_CRITIC': _hip.NET_CRITIC, 'MPNN_ACT_BN_MOVING': _hip.ACT_BN_MOVING,
              'MPNN_SLAB_ITEM': _hip.SLAB_ITEM, 'MPNN_LIN_KSLICES': _hip.LIN_KSLICES, 'MPNN_SEG_INTS': _hip.SEG_INTS,
              'MPNN_PREFIX_MAX': _hip.PREFIX_MAX, 'MPNN_LIN_RSPLIT': _hip.LIN_RSPLIT, 'MPNN_LIN_RS_TILE': _hip.LIN_RS_TILE}.get(name)
        if py is not None:
            assert py == int(val), name


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import arch_and_hypers as A
    from lib._hip import HipError
    net = A.sr_chain(1)((32, 32, 3), (10,))
    with pytest.raises(HipError):
        net.train.run({net.x0: np.zeros((2, 32, 32, 3)), net.y: np.zeros((2, 10)), net.mode: 'tr'})


def test_operator_surface_and_unicode_keywords():
    import arch_and_hypers as A
    from lib import layer_types as L, net_types as N
    for name in ['BatchNorm', 'Chain', 'CrossEntropyError', 'LinTrans', 'MultiscaleBatchNorm', 'MultiscaleConvMax',
                 'MultiscaleLLN', 'MultiscaleRect', 'Rect', 'Select', 'Softmax', 'ToPyramid', 'Conv', 'MaxPool',
                 'GlobalMaxPool', 'NoOp', 'Dropout', 'SquaredError', 'Layer']:
        assert hasattr(L, name)
    net = A.cr_chain(k_cpt=1e-9, optimistic=True)((32, 32, 3), (10,))
    assert isinstance(net, N.CriticNet) and net.hypers.optimistic and net.hypers.τ == 0.01
    # keyword identifiers are NFKC-normalised by Python; string lookups must agree (lib/_plan.py:_attr)
    from lib._plan import _attr
    assert _attr(net.hypers, 'ϵ') == 1e-6 and _attr(net.hypers, 'α_cpt') == 1e7
    blk = [ℓ for ℓ in net.layers if ℓ.name == 'ReConvMax'][0]
    conv = blk.comps[0]
    assert conv.params.w_horz_0.shape == (3, 3, 3, 16) and conv.params.w_vert_2.shape == (3, 3, 16, 16)
    assert blk.router.comps[-1].params.w.init == ('normal', 0.0)          # σ_w = 0: routers start at zero
    assert [s.shape for s in blk.x] == [(32, 32, 16), (16, 16, 16), (8, 8, 16), (4, 4, 16)]
    dyn = A.ac_chain(dyn_k_cpt=True)((32, 32, 3), (10,))
    r0 = [ℓ for ℓ in dyn.layers if ℓ.router][0].router
    assert r0.comps[1].params.w.shape == (257, 16)                         # flatten(4x4x16) + k_cpt column
    with pytest.raises(NotImplementedError):
        L.MultiscaleLLN().link([L.Sym((4, 4, 3))], None, 'tr')
    # MultiscaleBatchNorm accepts d / ϵ and DISCARDS them: every scale gets a BatchNorm() with the default hypers, as in the
    # reference (layer_types.py:246).  Rounds 1-5 forwarded them.
    ms = L.MultiscaleBatchNorm(d=0.5, ϵ=1e-3)
    ms.link([L.Sym((8, 8, 16)), L.Sym((4, 4, 16))], None, 'tr')
    assert [(c.hypers.d, _attr(c.hypers, 'ϵ')) for c in ms.comps] == [(0.9, 1e-6)] * 2 and ms.hypers.d == 0.5


@pytest.mark.skipif(not os.path.exists('/root/reference/scripts/arch_and_hypers.py'), reason='reference not mounted')
def test_reference_spec_file_drops_in_unchanged():
    """The reference's own arch_and_hypers.py, imported against THIS lib/ (same module names)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_arch', '/root/reference/scripts/arch_and_hypers.py')
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    import arch_and_hypers as A
    for mk_ref, mk in ((ref.sr_chain(5), A.sr_chain(5)), (ref.ac_chain(k_cpt=2e-9), A.ac_chain(k_cpt=2e-9)),
                       (ref.cr_chain(k_cpt=0.0, use_cls_err=True), A.cr_chain(k_cpt=0.0, use_cls_err=True))):
        a, b = mk_ref((32, 32, 3), (10,)), mk((32, 32, 3), (10,))
        assert type(a).__name__ == type(b).__name__ and vars(a.hypers) == vars(b.hypers)
        la, lb = list(a.layers), list(b.layers)
        assert [ℓ.name for ℓ in la] == [ℓ.name for ℓ in lb] and [ℓ.n_ops for ℓ in la] == [ℓ.n_ops for ℓ in lb]
        assert [(p.name, p.shape, p.init) for p in a._all_params] == [(p.name, p.shape, p.init) for p in b._all_params]
    assert (ref.arch, ref.k_cpts, ref.n_iter) == (A.arch, A.k_cpts, A.n_iter)


def test_train_nets_experiment_keys():
    import runpy
    ns = runpy.run_path(os.path.join(ROOT, 'multipath-nn_amd', 'train-nets'), run_name='not_main')
    keys = set(ns['experiments'])
    assert {'hybrid-sr', 'hybrid-ac', 'hybrid-ac-nokdec', 'hybrid-ac-notalr', 'hybrid-ac-tree', 'hybrid-cr',
            'hybrid-cr-opt', 'hybrid-cr-clserr', 'hybrid-cr-notalr', 'cifar2-sr', 'cifar2-ac', 'cifar5-sr',
            'cifar5-ac', 'cifar10-sr', 'cifar10-ac'} <= keys            # scripts/train-nets:28-88
    assert {'cifar10-cr', 'mnist-sr'} <= keys                             # BASELINE.json configs
    assert len(ns['experiments']['cifar10-ac'].nets) == 8 and len(ns['experiments']['cifar10-sr'].nets) == 8


def test_isa_has_no_uncovered_mfma_result_reads():
    """Every read of an MFMA accumulator by a VALU copy (v_accvgpr_read / v_accvgpr_mov) is separated
    from the last MFMA THAT WROTE THAT REGISTER by an s_nop or >= 11 vector-instruction slots
    (tools/scan_mfma_hazard.py; DESIGN.md section 3, "MFMA -> AGPR-copy hazard"), in every kernel of
    the library.  The scanner itself is checked on a synthetic listing first."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import scan_mfma_hazard as S
    with tempfile.TemporaryDirectory() as tmp:
        syn = os.path.join(tmp, 'syn.s')
        open(syn, 'w').write('_Z3badv:\n\tv_mfma_f32_16x16x4_f32 a[0:3], v1, v2, a[0:3]\n\ts_add_u32 s0, s0, 1\n\ts_cmp_lt_u32 s0, s1\n'
                             '\tv_accvgpr_read_b32 v5, a3\n'
                             '_Z4goodv:\n\tv_mfma_f32_16x16x4_f32 a[0:3], v1, v2, a[0:3]\n\tv_mfma_f32_16x16x4_f32 a[4:7], v1, v2, a[4:7]\n'
                             '\tv_mfma_f32_16x16x4_f32 a[8:11], v1, v2, a[8:11]\n\tv_accvgpr_read_b32 v5, a3\n'
                             '_Z4nop_v:\n\tv_mfma_f32_16x16x4_f32 a[0:3], v1, v2, a[0:3]\n\ts_nop 15\n\tv_accvgpr_read_b32 v5, a0\n')
        sites = S.scan(syn)
        assert [k for k, *_ in sites] == ['_Z3badv'], sites
    if shutil.which('hipcc') is None:
        pytest.skip('hipcc not available')
    import glob
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(S.CSRC, '*.hip'))):
            name = os.path.basename(src)
            out = os.path.join(tmp, name + '.s')
            subprocess.check_call(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                                   '-munsafe-fp-atomics', '-mllvm', '-amdgpu-kernarg-preload-count=16', '--cuda-device-only', '-S', src, '-o', out],
                                  cwd=S.CSRC, stderr=subprocess.DEVNULL)
            bad += [(name,) + site for site in S.scan(out)]
    assert not bad, bad[:5]


def test_isa_ticket_follows_store_acknowledgement():
    """lin.hip's cross-workgroup hand-offs (K-slices of lin_fwd_k, row groups of lin_bwd_k): between the last
    write-through (sc0 sc1) store of a partial tile and the ticket atomic every wave executes
    `s_waitcnt vmcnt(0)` and then the workgroup barrier -- otherwise the ticket can become visible before
    another wave's partials (MI355X_MICROARCH.md, "Valid forms")."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which('hipcc') is None:
        pytest.skip('hipcc not available')
    csrc = os.path.join(ROOT, 'multipath-nn_amd', 'csrc')
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'lin.s')
        subprocess.check_call(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                               '-munsafe-fp-atomics', '-mllvm', '-amdgpu-kernarg-preload-count=16', '--cuda-device-only', '-S', os.path.join(csrc, 'lin.hip'), '-o', out],
                              cwd=csrc, stderr=subprocess.DEVNULL)
        lines = [l.strip() for l in open(out)]
    tickets = [k for k, l in enumerate(lines) if l.split()[:1] in (['flat_atomic_add'], ['global_atomic_add']) and 'sc0 sc1' in l]
    assert len(tickets) >= 2, tickets                       # lin_fwd_k<4, true> and lin_bwd_k<..., true>
    for k in tickets:
        j = k
        while j > 0 and not ('_store_dword' in lines[j] and 'sc0 sc1' in lines[j]):
            j -= 1
        assert j > 0, 'ticket without a preceding write-through store'
        between = lines[j + 1:k]
        w = [i for i, l in enumerate(between) if l.startswith('s_waitcnt') and 'vmcnt(0)' in l]
        b = [i for i, l in enumerate(between) if l.startswith('s_barrier')]
        assert w and b and w[0] < b[-1], (lines[j], between)


def test_slab_item_size_keeps_one_load_batch_per_thread():
    """The planner's item size for mpnn_slab_reduce: a power of two in [64, MPNN_SLAB_ITEM] such that the
    kernel's slab groups (256 threads / (item / 4) element quads, at most 16) leave a thread <= 16 slabs
    wherever that is possible at all (up to 256 slabs)."""
    from lib import _hip
    for split in list(range(1, 70)) + [127, 128, 129, 255, 256]:
        item = _hip.slab_item_size(split)
        assert 64 <= item <= _hip.SLAB_ITEM and item & (item - 1) == 0
        groups = min(16, 256 // (item // 4))
        assert -(-split // groups) <= 16, (split, item, groups)
        if item < _hip.SLAB_ITEM:                      # never smaller than necessary
            g2 = min(16, 256 // (2 * item // 4))
            assert -(-split // g2) > 16, (split, item)
    assert _hip.slab_item_size(4096) == 64


def test_bench_launcher_spawns_ranks_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset becomes a launcher: torch.distributed.run with N ranks
    on 127.0.0.1, the original arguments passed through, the children's exit code returned -- and no HIP call
    in the parent (torch.cuda stays uninitialised)."""
    import importlib
    import subprocess
    import torch
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env

        class R:
            returncode = 7
        return R()
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    assert bench.main() == 7
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '4', '--steps', '3', '--warmup', '1'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert not torch.cuda.is_initialized()


def test_fast_augmentation_draws_replay_numpy_stream_exactly():
    """lib/data.py:_draw_augmentation_fast (the library's host function mpnn_draw_augmentation over raw MT19937 words)
    against the per-call numpy loop that mirrors scripts/lib/data.py:24-34: same (j, flip, du, dv) for every sample AND
    the same position of numpy's global stream afterwards, for set sizes on both sides of a power of two, shift ranges
    incl. 0, all / some / no symmetric classes, a one-image set (randint(0, 1) consumes no word)."""
    import numpy.random as rand
    from lib.data import _draw_augmentation, _draw_augmentation_fast, _sym_of_sources
    cases = [(0, 50000, 4, 10), (1, 60000, 4, 5), (2, 4096, 4, 10), (3, 1, 4, 0), (4, 65536, 0, 3), (5, 1000, 2, 7),
             (6, 50000, 4, 0), (7, 1, 0, 0), (8, 1, 0, 10), (9, 3, 1, 10), (10, 65537, 7, 9)]
    for seed, n_src, r, nsym in cases:
        g = np.random.default_rng(seed)
        y = np.eye(10)[g.integers(0, 10, n_src)]
        m_sym = np.zeros(10, bool); m_sym[:nsym] = True
        sym = _sym_of_sources(y, m_sym)
        su8 = np.asarray(sym, np.uint8)
        rand.seed(seed)
        ref = [_draw_augmentation(37, n_src, y, m_sym, r, sym) for _ in range(3)]
        tail_ref = rand.rand(3)
        for all_sym in ([False, True] if su8.all() else [False]):
            rand.seed(seed)
            out = [_draw_augmentation_fast(37, n_src, su8, r, all_sym=all_sym).copy() for _ in range(3)]
            tail = rand.rand(3)
            for (j, f, sh), o in zip(ref, out):
                assert np.array_equal(o[:, 0], j) and np.array_equal(o[:, 1].astype(bool), f) and np.array_equal(o[:, 2:], sh), (seed,)
            assert np.array_equal(tail, tail_ref), ('stream position', seed)


def test_fast_draws_reproduce_the_reference_fixtures():
    """The fixtures produced by the reference's own data.py (tests/golden/data_aug_golden.npz: seed, batch size, shift
    range and the batch the REFERENCE returned): the records of the fast path, pushed through the vectorised pixel
    assembly, give the reference's batches bit for bit."""
    import numpy.random as rand
    from lib.data import augmented_batch, _draw_augmentation_fast, _sym_of_sources
    G = np.load(os.path.join(ROOT, 'tests', 'golden', 'data_aug_golden.npz'))
    x0, y, m_sym = G['x0'], G['y'], G['m_sym']
    su8 = np.asarray(_sym_of_sources(y, m_sym), np.uint8)
    for k in range(3):
        seed, n, r = (int(v) for v in G['case%d_args' % k])
        rand.seed(seed)
        rec = _draw_augmentation_fast(n, len(x0), su8, r)
        xb, yb = augmented_batch(x0, y, n, m_sym, r, draws=rec)
        assert np.array_equal(xb, G['case%d_x' % k]) and np.array_equal(yb, G['case%d_y' % k]), k


def test_host_only_entry_points_without_a_gpu():
    """Entry points that are pure host state / validation run without a device: the limits of the any-width exit kernels
    (mpnn_exit_gen_check) and the compute-unit reservation of the persistent grids (mpnn_set_reserved_cus)."""
    from lib import _hip
    lib = _hip.load()
    ok = lib.mpnn_exit_gen_check
    assert ok(128, 2048, 100, 32, 24, 3) == 0 and ok(256, 4096, 1024, 256, 256, 4) == 0 and ok(16, 256, 0, 0, 0, 0) == 0
    assert ok(257, 2048, 10, 16, 16, 2) == _hip.E_SHAPE          # channels
    assert ok(128, 8192, 10, 16, 16, 2) == 0 and ok(128, 131072, 10, 16, 16, 2) == _hip.E_SHAPE      # features of the exit's input map
    assert ok(128, 2048, 1025, 16, 16, 2) == _hip.E_SHAPE        # classes
    assert ok(128, 2048, 10, 300, 16, 2) == _hip.E_SHAPE and ok(128, 2048, 10, 16, 300, 2) == _hip.E_SHAPE
    assert ok(128, 2048, 10, 16, 16, 5) == _hip.E_SHAPE and ok(128, 2048, 10, 16, 16, 1) == _hip.E_SHAPE      # sinks
    assert ok(48, 100, 10, 0, 0, 0) == _hip.E_SHAPE              # K not a multiple of C
    assert lib.mpnn_set_reserved_cus(-1) == 0                    # query
    assert lib.mpnn_set_reserved_cus(24) == 0 and lib.mpnn_set_reserved_cus(-1) == 24
    assert lib.mpnn_set_reserved_cus(0) == 24 and lib.mpnn_set_reserved_cus(-1) == 0


def test_draw_stream_is_numpys_legacy_stream():
    """lib.data.DrawStream (mpnn_draw_augmentation_mt: MT19937 stepped inside the library) against the reference's call
    sequence on numpy's global stream (scripts/lib/data.py:24-34): the same draws, batch after batch, and the same stream
    position afterwards -- also across the generator's 624-word refills; serial_positions hands net q the stream of the
    serial experiment loop after q nets of `iters` batches."""
    import numpy as np
    from lib import data as D
    ds = D.Dataset.synthetic(n_tr=300, n_ts=10, seed=1)
    ds.m_sym = np.array([1, 0, 1, 1, 0, 0, 1, 0, 1, 1], bool)
    sym = np.ascontiguousarray(D._sym_of_sources(ds.y_tr, ds.m_sym), dtype=np.uint8)
    for seed, n, r in ((0, 33, 4), (7, 128, 4), (3, 5, 0), (9, 64, 1)):
        np.random.seed(seed)
        s = D.DrawStream(seed)
        for b in range(6):
            j, flip, sh = D._draw_augmentation(n, 300, ds.y_tr, ds.m_sym, r)
            out = s.draw(n, 300, sym, r, np.empty((n, 4), np.int32))
            assert np.array_equal(out[:, 0], j) and np.array_equal(out[:, 1].astype(bool), flip) and np.array_equal(out[:, 2:], sh), (seed, b)
        rs = np.random.RandomState()
        rs.set_state(s.state())
        assert np.array_equal(rs.randint(0, 2 ** 32, 700, dtype=np.uint32), np.random.randint(0, 2 ** 32, 700, dtype=np.uint32))
    # several batches per call == one per call; skipping == drawing and discarding
    a, b, c = D.DrawStream(5), D.DrawStream(5), D.DrawStream(5)
    many = a.draw(17, 300, sym, 3, np.empty((4, 17, 4), np.int32), batches=4)
    for k in range(4):
        assert np.array_equal(many[k], b.draw(17, 300, sym, 3, np.empty((17, 4), np.int32)))
    c.skip(4, 17, 300, sym, 3)
    assert np.array_equal(c.key, a.key) and c.pos.value == a.pos.value
    # positions of the serial loop
    ps = ds.serial_positions([0, 2, 3], 10, n=17, r_shift=3, seed=5)
    np.random.seed(5)
    for q in range(4):
        for it in range(10):
            j, flip, sh = D._draw_augmentation(17, 300, ds.y_tr, ds.m_sym, 3)
            if q in (0, 2, 3):
                out = ps[[0, 2, 3].index(q)].draw(17, 300, sym, 3, np.empty((17, 4), np.int32))
                assert np.array_equal(out[:, 0], j) and np.array_equal(out[:, 1].astype(bool), flip) and np.array_equal(out[:, 2:], sh), (q, it)


def test_ev_prefix_walk_refuses_bad_records():
    """mpnn_ev_prefix_walk validates the host record before anything is launched (no GPU needed to be refused)."""
    import ctypes as C
    from lib import _hip
    lib = _hip.load()
    a = _hip.EvPrefixArgs()
    assert lib.mpnn_ev_prefix_walk(None, None, None) == _hip.E_ARG
    a.n = 0
    assert lib.mpnn_ev_prefix_walk(C.byref(a), 1, None) == 0                  # (nothing to do)
    a.n, a.count = 16, 0
    assert lib.mpnn_ev_prefix_walk(C.byref(a), 1, None) == _hip.E_SHAPE
    a.count = _hip.PREFIX_MAX + 1
    assert lib.mpnn_ev_prefix_walk(C.byref(a), 1, None) == _hip.E_SHAPE
    a.count = 2
    a.parent[0], a.parent[1] = -1, 1                                          # a record below itself
    assert lib.mpnn_ev_prefix_walk(C.byref(a), 1, None) == _hip.E_ARG
    a.parent[1], a.parent_sink[1], a.n_sinks[0] = 0, 2, 2                      # sink 2 of a two-way switch
    a.r[0], a.r_stride[0] = 64, 2
    assert lib.mpnn_ev_prefix_walk(C.byref(a), 1, None) == _hip.E_ARG
    a.parent_sink[1], a.n_sinks[0] = 1, _hip.MAX_SINKS + 1
    assert lib.mpnn_ev_prefix_walk(C.byref(a), 1, None) == _hip.E_SHAPE
    a.n_sinks[0], a.r_stride[0] = 2, 1                                         # rows narrower than the sinks
    assert lib.mpnn_ev_prefix_walk(C.byref(a), 1, None) == _hip.E_ARG
    a.r_stride[0], a.n_front = 2, 1
    a.front_parent[0], a.front_sink[0] = 1, 0                                  # a list below a node without a switch
    assert lib.mpnn_ev_prefix_walk(C.byref(a), 1, None) == _hip.E_ARG
