"""Ablation of the conv bodies (skip the MFMAs / the staging / the epilogue of a unit, MPNN_CONV_DBG bits 1 / 2 / 4) on a few
launches of the training step.  Needs a library built with the ablation branches compiled in:
    make -C multipath-nn_amd/csrc EXTRA=-DMPNN_ABLATE      (production builds fold them away)
"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import torch, arch_and_hypers as A
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine(); n = 128
eng.x0[:n].uniform_(); eng.y[:n].zero_(); eng.y[:n, 0] = 1
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.0, net.τ: 1.0}
net.train.run(feed)
eng.multi_stream = True     # separate launches (they honour MPNN_CONV_DBG)
want = {'h4 128+0->128', 'h8 64+0->64', 'h32 16+0->16', 'h4 64+64->64', 'h16 32+0->32'}
for dbg in (0, 1, 2, 4, 3, 7):
    os.environ['MPNN_CONV_DBG'] = str(dbg)
    ops = eng.time_ops('tr', n, reps=20)
    print('dbg', dbg, ' '.join('%s=%.1f' % (o[1].replace(' ', ''), o[3] * 1e3) for o in ops if o[0] == 'msconv_fwd' and o[1] in want))
os.environ['MPNN_CONV_DBG'] = '0'
