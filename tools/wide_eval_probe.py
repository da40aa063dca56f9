import os, sys
ROOT='/root/repo'
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, time, arch_and_hypers as A
from lib.net_types import ActorNet
from test_net_parity import _wide_chain
for name, widths, n_cls in (('10 cls 16-16', (16,16), 10), ('100 cls 32-32', (32,32), 100)):
    net = _wide_chain(ActorNet, widths, n_blocks=8, k_cpt=1.6e-8)((32,32,3),(n_cls,))
    eng = net.engine(); eng.init_params(5)
    for nb in (128, 4096):
        x = torch.rand(nb,32,32,3,device='cuda'); y = torch.zeros(nb,n_cls,device='cuda'); y[:,0]=1
        feed={net.x0:x, net.y:y}
        for _ in range(3): net.eval(feed)
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(10): net.eval(feed)
        torch.cuda.synchronize(); ms=(time.perf_counter()-t)/10*1e3
        print(name, 'dense eval batch', nb, '%.3f ms' % ms, flush=True)
        if nb == 4096:
            prog = eng.program('ev', nb)
            st = torch.cuda.current_stream()
            for op in prog['fwd']:
                if op.what in ('exit_ev', 'route'):
                    eng._begin(False)
                    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
                    op(st.cuda_stream); e0.record(st); op(st.cuda_stream); e1.record(st); torch.cuda.synchronize()
                    print('    ', op.what, '%.1f us' % (e0.elapsed_time(e1)*1e3))
