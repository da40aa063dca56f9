"""The 16 -> 16 channel conv on a big map at an evaluation batch: general body (mpnn_msconv_fwd) against the
wave-per-strip body (a one-member mpnn_msconv_fwd_group; MPNN_STRIP=0 switches it off), moving-average BatchNorm on load.
    python tools/strip_probe.py [n] [H]
"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from lib import _hip
import hiputil as U
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lib = _hip.load()
rng = np.random.default_rng(0)
ws = [rng.standard_normal((3, 3, 16, 16)).astype(np.float32)]
fp, _ = U.pack_weights(ws, want_bwd=False)
x = torch.randn(n, H, H, 16, device='cuda')
out = torch.empty(n, H, H, 16, device='cuda')
pool = torch.empty(n, H // 2, H // 2, 16, device='cuda')
bias = torch.zeros(16, device='cuda')
bn = dict(sum=None, gamma=torch.ones(16, device='cuda'), beta=torch.zeros(16, device='cuda'), m_avg=torch.zeros(16, device='cuda'),
          v_avg=torch.ones(16, device='cuda'), eps=1e-6)
a = _hip.ConvFwdArgs()
a.a = _hip.act(x, 16, _hip.ACT_BN_MOVING, 0, bn, n * H * H)
a.wa_pack, a.bias, a.out, a.pool_out = fp[0].data_ptr(), bias.data_ptr(), out.data_ptr(), pool.data_ptr()
a.n, a.H, a.W, a.Cout, a.out_nslot = n, H, H, 16, 8
arr = (_hip.ConvFwdArgs * 1)(a)
tab = _hip.to_device_table([a], 'cuda')
st = torch.cuda.current_stream()
fl = 2.0 * n * H * H * 9 * 16 * 16
for name, f in (('general body', lambda: lib.mpnn_msconv_fwd(C.byref(a), st.cuda_stream)),
                ('group launch', lambda: lib.mpnn_msconv_fwd_group(arr, tab.data_ptr(), 1, st.cuda_stream))):
    for _ in range(3): _hip.check(f(), name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10): f()
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print('%-13s h%d 16->16, %d images: %8.1f us  %6.1f TFLOP/s' % (name, H, n, us, fl / us / 1e6))
