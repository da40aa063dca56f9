import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import faulthandler; faulthandler.enable()
import torch, arch_and_hypers as A
what, nops = sys.argv[1], int(sys.argv[2])
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine(); n = 32
eng.x0[:n].uniform_(); eng.y[:n].zero_(); eng.y[:n, 0] = 1
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.0, net.τ: 1.0}
net.train.run(feed); torch.cuda.synchronize()
prog = eng.program('tr', n)
eng.multi_stream = True
ops = list(prog[what])
if nops > 0:
    # keep fork, first nops real ops, then a join
    real = [o for o in ops if o.what not in ('fork', 'join')][:nops]
    ops = [ops[0]] + real + [o for o in ops if o.what == 'join'][:1]
print([ (o.what, o.stream) for o in ops][:12], len(ops), flush=True)
eng._launch(ops); torch.cuda.synchronize(); print('eager ok', flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng._launch(ops)
print('captured', flush=True)
g.replay(); torch.cuda.synchronize(); print('replayed ok')
