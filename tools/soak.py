"""Soak: thousands of training steps on random batches (finite parameters, loss goes down on a fixed
batch), then the same-state repeatability check many times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import numpy as np, torch, arch_and_hypers as A

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
net = A.ac_chain(k_cpt=1.6e-8, seed=3)((32, 32, 3), (10,)); eng = net.engine()
g = torch.Generator(device='cuda').manual_seed(0)
xs = torch.rand((8, 128, 32, 32, 3), device='cuda', generator=g)
ys = torch.eye(10, device='cuda')[torch.randint(0, 10, (8, 128), device='cuda', generator=g)]
t0 = time.time()
for t in range(n_steps):
    k = t % 8
    net.train.run({net.x0: xs[k], net.y: ys[k], net.mode: 'tr', net.λ_lrn: A.λ_lrn(t), net.τ: A.τ_ds(t)})
    if t % 1000 == 999:
        torch.cuda.synchronize()
        assert torch.isfinite(eng.P).all() and torch.isfinite(eng.S).all(), t
        st = net.state()
        print('step %d  acc %.3f  finite ok  (%.1f s)' % (t + 1, float(st[(net, 'acc')].mean()), time.time() - t0), flush=True)
# repeatability from the trained state
P0, A0, S0 = eng.P.clone(), eng.A.clone(), eng.S.clone()
feed = {net.x0: xs[0], net.y: ys[0], net.mode: 'tr', net.λ_lrn: 0.01, net.τ: 1.0}
ref = None
worst = 0.0
for rep in range(200):
    eng.P.copy_(P0); eng.A.copy_(A0); eng.S.copy_(S0); eng.invalidate_packs()
    net.train.run(feed); torch.cuda.synchronize()
    s = [b.s[i].clone() for b in eng.blocks for i in range(b.L)]
    if ref is None:
        ref = (s, eng.G.clone())
    else:
        assert all(torch.equal(a, b) for a, b in zip(s, ref[0])), 'forward sums differ at repetition %d' % rep
        worst = max(worst, float((eng.G - ref[1]).abs().max() / ref[1].abs().max()))
print('200 repetitions: forward sums bit-identical, max relative gradient difference %.2e' % worst)
