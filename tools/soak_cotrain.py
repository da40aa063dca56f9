"""Soak of the round-5 paths: (a) K-step graphs (Engine.run_steps, 4 steps per replay, schedules changing every step) and
(b) a co-trained group of 8 nets (lib/_co.py), thousands of steps on a few fixed batches: parameters stay finite, the
training accuracy on those batches goes up, and from one saved state the same joint step gives the same result every
time (forward sums bit-identical, gradients to the fp32-atomic order of the exit path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import numpy as np, torch, arch_and_hypers as A
from lib._co import CoTrainer

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g = torch.Generator(device='cuda').manual_seed(0)
xs = torch.rand((8, 128, 32, 32, 3), device='cuda', generator=g)
ys = torch.eye(10, device='cuda')[torch.randint(0, 10, (8, 128), device='cuda', generator=g)]

# ---- (a) K-step graphs
net = A.ac_chain(k_cpt=1.6e-8, seed=3)((32, 32, 3), (10,)); eng = net.engine()
eng._ensure_capacity(128)
t0 = time.time()
for t in range(0, n_steps, 4):
    k = (t // 4) % 8
    eng.x0[:128].copy_(xs[k]); eng.y[:128].copy_(ys[k])
    net.train.run_steps([{net.x0: eng.x0[:128], net.y: eng.y[:128], net.mode: 'tr', net.λ_lrn: A.λ_lrn(t + j), net.τ: A.τ_ds(t + j)}
                         for j in range(4)])
    if (t + 4) % 1000 == 0:
        torch.cuda.synchronize()
        assert torch.isfinite(eng.P).all() and torch.isfinite(eng.S).all(), t
        net.eval({net.x0: xs[0], net.y: ys[0]})
        print('run_steps: step %d  acc on batch 0 %.3f  finite ok  (%.1f s)' % (t + 4, float(net.state()[(net, 'acc')].mean()), time.time() - t0), flush=True)

# ---- (b) co-trained group
nets = [A.ac_chain(k_cpt=k, seed=10 + i)((32, 32, 3), (10,)) for i, k in enumerate(A.k_cpts)]
engs = [m.engine() for m in nets]
for e in engs: e._ensure_capacity(128)
co = CoTrainer(nets)
t0 = time.time()
for t in range(n_steps):
    feeds = []
    for i, (m, e) in enumerate(zip(nets, engs)):
        k = (t + i) % 8
        feeds.append({m.x0: xs[k], m.y: ys[k], m.mode: 'tr', m.λ_lrn: A.λ_lrn(t), m.τ: A.τ_ds(t)})
    co.run(feeds)
    if (t + 1) % 1000 == 0:
        torch.cuda.synchronize()
        accs = []
        for m, e in zip(nets, engs):
            assert torch.isfinite(e.P).all() and torch.isfinite(e.S).all(), t
            m.eval({m.x0: xs[0], m.y: ys[0]})
            accs.append(float(m.state()[(m, 'acc')].mean()))
        print('co-trained: step %d  acc on batch 0 per net %s  finite ok  (%.1f s)' % (t + 1, ' '.join('%.2f' % a for a in accs), time.time() - t0), flush=True)
saved = [(e.P.clone(), e.A.clone(), e.S.clone()) for e in engs]
feeds = [{m.x0: xs[i], m.y: ys[i], m.mode: 'tr', m.λ_lrn: 0.01, m.τ: 1.0} for i, m in enumerate(nets)]
ref, worst = None, 0.0
for rep in range(100):
    for e, (P, A_, S) in zip(engs, saved):
        e.P.copy_(P); e.A.copy_(A_); e.S.copy_(S); e.invalidate_packs()
    co.run(feeds); torch.cuda.synchronize()
    s = [b.s[i].clone() for e in engs for b in e.blocks for i in range(b.L)]
    G = [e.G.clone() for e in engs]
    if ref is None:
        ref = (s, G)
    else:
        assert all(torch.equal(a, b) for a, b in zip(s, ref[0])), 'forward sums differ at repetition %d' % rep
        worst = max(worst, max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(G, ref[1])))
print('100 repetitions of the joint step: forward sums of all 8 nets bit-identical, max relative gradient difference %.2e' % worst)
