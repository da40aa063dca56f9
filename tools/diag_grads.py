import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/multipath-nn_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, arch_and_hypers as A
from oracle.ref_net import RefNet
from test_net_parity import batch, perturb_routers
for name, mk, n, tau in [('ac', A.ac_chain(k_cpt=1.6e-8), 16, 1.0), ('cr', A.cr_chain(k_cpt=8e-9), 16, 0.1)]:
    net = mk((32,32,3),(10,)); eng=net.engine()
    perturb_routers(net)
    ref = RefNet(net); ref.load_params()
    for t in range(2):
        x0,y = batch(n,3,seed=t)
        feed={net.x0:x0, net.y:y, net.mode:'tr', net.λ_lrn:0.05, net.τ:tau}
        before = {id(p): p.data.clone() for p in net._all_params if p.trainable}
        net.train.run(feed); torch.cuda.synchronize()
        res = ref.train_step(x0,y,0.05,τ=tau)
        R=lambda l: res['out'][id(l)]
        print('==',name,'step',t)
        for i,l in enumerate(net.layers):
            e=np.abs(l.p_tr.cpu().numpy()-R(l)['p_tr'].detach().numpy()).max()
            if l.router is not None:
                rx=R(l.router)['x'].detach().numpy(); er=np.abs(l.router.x.cpu().numpy()-rx).max()
                print(i,l.name,'p_tr err %.2e  r err %.2e (max|r| %.2f)'%(e,er,np.abs(rx).max()))
            elif e>1e-5: print(i,l.name,'p_tr err %.2e'%e)
        rows=[]
        for p in net._all_params:
            if not p.trainable: continue
            g_ref=res['grads'][id(p)].numpy().reshape(-1); g=p.grad.cpu().numpy()
            if p.l2:
                g = g + 2*p.l2*float(eng.nodes[p.node].layer.p_tr.mean())*before[id(p)].cpu().numpy()
            sc=np.abs(g_ref).max(); e=np.abs(g-g_ref).max()
            rows.append((e/(sc+1e-30), p.node, p.is_router, p.owner.name, p.name, e, sc))
        rows.sort(key=lambda r: -r[0])
        print('n rows', len(rows), flush=True)
        k=0
        for r in rows:
            if r[6]>1e-9 and k<14:
                k+=1; print('  rel %.2e node %d rtr %d %s.%s err %.2e scale %.2e'%r, flush=True)
