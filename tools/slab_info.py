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
# the slab item table: 6 ints per item (src_off, dst_off, count <= 1024, n_split >= 1, stride > 0, 0) -- picked out of the
# plan's device tables by that signature (the optimizer's tables have 12 or 4 ints per row)
cands = [t for t in eng._keep if isinstance(t, torch.Tensor) and t.dtype == torch.int32 and t.dim() == 1 and t.numel() % 6 == 0 and t.numel() > 600]
def is_slab(t):
    a = t.cpu().numpy().reshape(-1, 6)
    return a[:, 5].max() == 0 and a[:, 3].min() >= 1 and a[:, 4].min() > 0 and a[:, 2].max() <= 1024 and a[:, 2].min() >= 1
tab = [t for t in cands if is_slab(t)][-1]
t = tab.cpu().numpy().reshape(-1, 6)
read = int((t[:, 2].astype('int64') * t[:, 3]).sum()) * 4
print('items %d, elements %d, slab bytes read %.1f MB' % (len(t), int(t[:, 2].sum()), read / 1e6))
by = collections.OrderedDict()
for row in t:
    k = (int(row[3]), int(row[2]))
    by[k] = by.get(k, 0) + 1
for (ns, cnt), n in sorted(by.items()):
    print('  split %4d  item %4d elements: %5d items, %.2f MB' % (ns, cnt, n, ns * cnt * n * 4 / 1e6))
