import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/multipath-nn_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, arch_and_hypers as A
from oracle.ref_net import RefNet
from test_net_parity import batch
net = A.sr_chain(8)((32,32,3),(10,)); eng=net.engine()
x0,y = batch(8,3,seed=0)
ref=None; gs=[]
for rep in range(4):
    eng.init_params(1234)
    if ref is None:
        ref = RefNet(net); ref.load_params(); res = ref.train_step(x0,y,0.05)
    net.train.run({net.x0:x0, net.y:y, net.mode:'tr', net.λ_lrn:0.05}); torch.cuda.synchronize()
    g = eng.G[:eng.n_params].cpu().numpy().copy(); gs.append(g)
    rows=[]
    for p in net._all_params:
        if not p.trainable: continue
        gr=res['grads'][id(p)].numpy().reshape(-1); gp=g[p.offset:p.offset+p.size].copy()
        if p.l2: gp = gp + 2*p.l2*ref_before[id(p)] if False else gp
        sc=np.abs(gr).max()
        if sc>1e-9 and not p.l2: rows.append((np.abs(gp-gr).max()/sc, p.owner.name+'.'+p.name, p.node))
    rows.sort(key=lambda r:-r[0])
    print('rep',rep,'worst (no-L2 tensors):',['%.1e %s n%d'%r for r in rows[:4]], ' run-to-run max|dG| vs rep0: %.2e (max|G| %.2e)'%(np.abs(g-gs[0]).max(), np.abs(gs[0]).max()), flush=True)
# where do reps differ?
for rep in range(1,4):
    d=np.abs(gs[rep]-gs[0]); i=int(d.argmax())
    for p in net._all_params:
        if p.trainable and p.offset<=i<p.offset+p.size: print('rep',rep,'largest diff in',p.owner.name,p.name,'node',p.node,'%.2e'%d.max(), 'tensor max %.2e'%np.abs(gs[0][p.offset:p.offset+p.size]).max())
