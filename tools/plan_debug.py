import os, sys
os.environ['MPNN_PLAN_DEBUG'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd')]
import torch, arch_and_hypers as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
share = int(sys.argv[2]) if len(sys.argv) > 2 else 1
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
eng.ensure_capacity(n)
eng.co_share = share
print('== batch %d, co_share %d' % (n, share))
eng.program('tr', n)
