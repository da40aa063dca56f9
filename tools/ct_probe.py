"""Output-channel tile of the forward conv at evaluation batch sizes: single-op launches (mpnn_msconv_fwd honours
MPNN_CONV_CT = 16 / 32 / 64) of the deep layers of the chain at n images, moving-average BatchNorm on load.
    MPNN_CONV_CT=64 python tools/ct_probe.py 4096
"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from lib import _hip
import hiputil as U
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _hip.load()
rng = np.random.default_rng(0)
tot = 0.0
for H, Ca, Cv, Co in [(16, 32, 0, 32), (8, 32, 32, 32), (8, 32, 0, 64), (4, 32, 64, 64), (8, 64, 0, 64), (4, 64, 64, 64), (4, 64, 0, 128), (4, 128, 0, 128)]:
    ws = [rng.standard_normal((3, 3, Ca, Co)).astype(np.float32)] + ([rng.standard_normal((3, 3, Cv, Co)).astype(np.float32)] if Cv else [])
    fp, _ = U.pack_weights(ws, want_bwd=False)
    x = torch.randn(n, H, H, Ca, device='cuda')
    v = torch.randn(n, H, H, Cv, device='cuda') if Cv else None
    out = torch.empty(n, H, H, Co, device='cuda')
    bias = torch.zeros(Co, device='cuda')
    bn = dict(sum=None, gamma=torch.ones(Ca, device='cuda'), beta=torch.zeros(Ca, device='cuda'), m_avg=torch.zeros(Ca, device='cuda'),
              v_avg=torch.ones(Ca, device='cuda'), eps=1e-6)
    a = _hip.ConvFwdArgs()
    a.a = _hip.act(x, Ca, _hip.ACT_BN_MOVING, 0, bn, n * H * H)
    if Cv:
        a.v, a.Cv, a.wv_pack = v.data_ptr(), Cv, fp[1].data_ptr()
    a.wa_pack, a.bias, a.out = fp[0].data_ptr(), bias.data_ptr(), out.data_ptr()
    a.n, a.H, a.W, a.Cout, a.out_nslot = n, H, H, Co, 8
    st = torch.cuda.current_stream()
    for _ in range(3): _hip.check(lib.mpnn_msconv_fwd(C.byref(a), st.cuda_stream), 'fwd')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10): lib.mpnn_msconv_fwd(C.byref(a), st.cuda_stream)
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2.0 * n * H * H * 9 * Co * (Ca + Cv)
    tot += us
    print('h%-2d %3d+%-3d->%-3d  %8.1f us  %6.1f TFLOP/s' % (H, Ca, Cv, Co, us, fl / us / 1e6))
print('CT=%s total %.1f us' % (os.environ.get('MPNN_CONV_CT', 'default'), tot))
