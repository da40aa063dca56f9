import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import torch, arch_and_hypers as A
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine(); n = 128
eng.x0[:n].uniform_(); eng.y[:n].zero_(); eng.y[:n, 0] = 1
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.0, net.τ: 1.0}
net.train.run(feed)
ops = eng.time_ops('tr', n, reps=20)
agg = {}
for what, tag, fl, ms in ops:
    if what in ('msconv_fwd', 'dgrad_horz', 'dgrad_vert'):
        k = (what, tag.split()[0])
        agg[k] = agg.get(k, 0) + ms * 1e3
print(os.environ.get('MPNN_CONV_CT', 'default'), ' '.join('%s/%s=%.0f' % (k[0][:6], k[1], v) for k, v in sorted(agg.items())),
      'TOTAL %.0f' % sum(agg.values()), ' lin_bwd %.1f' % sum(o[3] * 1e3 for o in ops if o[0] == 'lin_bwd'))
