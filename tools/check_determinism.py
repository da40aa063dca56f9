import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import numpy as np, torch, arch_and_hypers as A
def fresh():
    net = A.ac_chain(k_cpt=1.6e-8, seed=5)((32, 32, 3), (10,)); return net, net.engine()
rng = np.random.default_rng(0)
x0 = rng.random((128, 32, 32, 3)).astype(np.float32); y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 128)]
feed = lambda net: {net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.τ: 1.0}
# (1) same engine, same state, two steps
net, eng = fresh()
for _ in range(3): net.train.run(feed(net))
P0, A0, S0 = eng.P.clone(), eng.A.clone(), eng.S.clone()
outs = []
for rep in range(3):
    eng.P.copy_(P0); eng.A.copy_(A0); eng.S.copy_(S0); eng.invalidate_packs()
    net.train.run(feed(net)); torch.cuda.synchronize()
    outs.append((eng.P.clone(), eng.G.clone()))
for r in (1, 2):
    print('same engine rep', r, 'max|dP| %.3e  max|dG| %.3e' % ((outs[r][0] - outs[0][0]).abs().max().item(), (outs[r][1] - outs[0][1]).abs().max().item()))
# (2) fresh engine loaded with the same state
net2, eng2 = fresh()
eng2.P.copy_(P0); eng2.A.copy_(A0); eng2.S.copy_(S0); eng2.invalidate_packs()
net2.train.run(feed(net2)); torch.cuda.synchronize()
d = (eng2.G - outs[0][1]).abs()
print('fresh engine: max|dP| %.3e  max|dG| %.3e' % ((eng2.P - outs[0][0]).abs().max().item(), d.max().item()))
i = int(d.argmax())
for p in net._all_params:
    if p.trainable and p.offset <= i < p.offset + p.size: print(' largest dG in', p.owner.name, p.name, 'node', p.node, 'scale of G there %.3e' % outs[0][1][p.offset:p.offset+p.size].abs().max().item())
