"""Output-channel tile width of the input-gradient bodies WHERE WORK IS PLENTIFUL: single mpnn_msconv_dgrad_horz /
mpnn_msconv_dgrad_vert launches at 1 024 and 4 096 images with 16-, 32- and 64-channel output tiles (MPNN_CONV_CT, read
once per process: one child process per width).  The level launches of the training step run the 16-channel bodies
(bwd_level_k.h); round 1's measurement that 16 beats 32 / 64 on the small maps was made at batch 128.

    python tools/dgrad_ct_probe.py            # table: us per launch and TFLOP/s per (shape, width)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

SHAPES = [  # (kind, H, C of g, C of the output)
    ('h', 4, 128, 128), ('h', 4, 64, 64), ('h', 4, 32, 32), ('h', 8, 64, 64), ('h', 8, 32, 32), ('h', 16, 32, 32),
    ('v', 4, 128, 64), ('v', 4, 64, 32), ('v', 8, 64, 32), ('v', 8, 32, 32),
]


def child(n):
    import numpy as np
    import torch
    import hiputil as hu
    from lib import _hip
    lib = _hip.load()
    rng = np.random.default_rng(0)
    for kind, H, cg, co in SHAPES:
        g = torch.randn(n, H, H, cg, device='cuda')
        w = (rng.standard_normal((3, 3, co, cg)) * 0.1).astype(np.float32)
        _, bw = hu.pack_weights([w])
        bn = dict(sum=torch.zeros(_hip.BN_SLOTS * 2 * co, device='cuda', dtype=torch.float64), gamma=torch.ones(co, device='cuda'),
                  beta=torch.zeros(co, device='cuda'), m_avg=torch.zeros(co, device='cuda'), v_avg=torch.ones(co, device='cuda'), eps=1e-6)
        bn['sum'][co:2 * co] = float(n * H * H) / 1.0
        if kind == 'h':
            s = torch.randn(n, H, H, co, device='cuda')
            out = torch.empty(n, H, H, co, device='cuda')
            red = torch.zeros(_hip.BN_SLOTS * 2 * co, device='cuda', dtype=torch.float64)
            ctx = hu.bn_ctx(s, co, bn, n * H * H)
            a = _hip.DgradHorzArgs()
            a.g = g.data_ptr(); a.Cg = cg; a.w_pack = bw[0].data_ptr(); a.out = out.data_ptr()
            a.n, a.H, a.W, a.Cout = n, H, H, co
            a.prev = C.pointer(ctx); a.red_out = red.data_ptr()
            fn = lib.mpnn_msconv_dgrad_horz
        else:
            s = torch.randn(n, 2 * H, 2 * H, co, device='cuda')
            out = torch.randn(n, 2 * H, 2 * H, co, device='cuda')
            redd = torch.zeros(_hip.BN_SLOTS * 2 * co, device='cuda', dtype=torch.float64)
            ctx = hu.bn_ctx(s, co, bn, n * 4 * H * H, red=redd)
            a = _hip.DgradVertArgs()
            a.g = g.data_ptr(); a.Cg = cg; a.w_pack = bw[0].data_ptr(); a.fine = C.pointer(ctx); a.fine_has_dz = 1
            a.dz_g_fine = out.data_ptr(); a.n, a.H, a.W, a.Cout = n, H, H, co
            fn = lib.mpnn_msconv_dgrad_vert
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            _hip.check(fn(C.byref(a), st), 'dgrad')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            fn(C.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        fl = 2.0 * n * H * H * 9 * cg * co
        print('%s h%-2d %3d->%-3d n %4d  %8.1f us  %6.1f TFLOP/s' % (kind, H, cg, co, n, us, fl / us * 1e-6), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(int(sys.argv[2]))
        sys.exit(0)
    for n in (1024, 4096):
        for ct in (16, 32, 64):
            print('== n %d, %d-channel output tiles' % (n, ct), flush=True)
            env = dict(os.environ, MPNN_CONV_CT=str(ct))
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(n)], env=env, check=False)
