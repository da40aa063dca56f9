"""Slab volume of the bench configuration: what mpnn_backward_finish reads.

    python tools/slab_info.py          (needs the GPU: the plan is built by the engine)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import collections
import torch, arch_and_hypers as A

net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
eng.program('tr', 128)
tab = [t for t in eng._keep if isinstance(t, torch.Tensor) and t.dtype == torch.int32 and t.dim() == 1 and t.numel() % 6 == 0 and t.numel() > 600][-1]
t = tab.cpu().numpy().reshape(-1, 6)
read = int((t[:, 2].astype('int64') * t[:, 3]).sum()) * 4
print('items %d, elements %d, slab bytes read %.1f MB' % (len(t), int(t[:, 2].sum()), read / 1e6))
by = collections.OrderedDict()
for row in t:
    k = (int(row[3]), int(row[2]))
    by[k] = by.get(k, 0) + 1
for (ns, cnt), n in sorted(by.items()):
    print('  split %4d  item %4d elements: %5d items, %.2f MB' % (ns, cnt, n, ns * cnt * n * 4 / 1e6))
