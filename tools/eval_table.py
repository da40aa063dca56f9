"""Per-launch table of the evaluation program at a statistics-pass batch (HIP events around every launch).
    python tools/eval_table.py [batch] [routed]        (default: the dense program)
"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import torch, numpy as np, arch_and_hypers as A
import bench
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
x, y = bench.synthetic(nb, 1, 'cuda:0')
eng._ensure_capacity(nb, train=False)
eng.x0[:nb].copy_(x); eng.y[:nb].copy_(y)
feed = {net.x0: eng.x0[:nb], net.y: eng.y[:nb]}
for _ in range(3): net.eval(feed)
routed = len(sys.argv) > 2 and sys.argv[2] == 'routed'
prog = eng.program('ev', nb, routed=routed)
st = torch.cuda.current_stream()
tot = [0.0] * len(prog['fwd'])
for rep in range(6):
    eng._begin(False)
    evs = []
    for op in prog['fwd']:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); op(st.cuda_stream); e1.record(st)
        evs.append((e0, e1))
    torch.cuda.synchronize()
    if rep:
        for k, (e0, e1) in enumerate(evs): tot[k] += e0.elapsed_time(e1) / 5
for op, t in zip(prog['fwd'], tot):
    print('%-10s %-52s %8.1f us  %6.1f TFLOP/s' % (op.what, op.tag[:52], t * 1e3, op.flops / (t * 1e-3) / 1e12 if op.flops else 0))
print('sum %.1f us = %.2f M img/s' % (sum(tot) * 1e3, nb / sum(tot) / 1e3))
