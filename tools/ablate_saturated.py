"""Upper bounds for "hide the staging under the MFMAs": the single-op conv launches of a training step at a SATURATING batch
(default 1 024 images) with parts of every unit skipped (MPNN_CONV_DBG bits: 1 = no MFMAs, 2 = no staging stores, 4 = no
epilogue).  Needs the ablation build:
    bash multipath-nn_amd/csrc/build_variant.sh ablate -DMPNN_ABLATE
    MPNN_HIP_LIB=multipath-nn_amd/libmpnn_hip_ablate.so python tools/ablate_saturated.py [batch] --all
Results are wrong by construction (work is skipped); only the times mean anything."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
# (MPNN_CONV_DBG is read ONCE per process by the launchers: one process per setting -- `--all` runs them as children)
import subprocess
if '--all' in sys.argv:
    tabs = {}
    for dbg in (0, 1, 2, 4, 6, 7):
        out = subprocess.check_output([sys.executable, __file__, str(n)], env=dict(os.environ, MPNN_CONV_DBG=str(dbg))).decode()
        tabs[dbg] = [ln.split('\t') for ln in out.splitlines() if ln.startswith('op\t')]
    print('batch %d; us per launch: full | no MFMAs | no staging | no epilogue | MFMAs only | nothing' % n)
    tot = {d: 0.0 for d in tabs}
    for k, row in enumerate(tabs[0]):
        print('%-16s %-30s %s' % (row[1], row[2][:30], ' '.join('%8.1f' % float(tabs[d][k][3]) for d in tabs)))
        for d in tabs:
            tot[d] += float(tabs[d][k][3])
    print('%-47s %s' % ('sum', ' '.join('%8.1f' % tot[d] for d in tabs)))
    sys.exit(0)
import torch, arch_and_hypers as A
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
eng._ensure_capacity(n)
eng.x0[:n].uniform_(); eng.y[:n].zero_(); eng.y[:n, 0] = 1
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.0, net.τ: 1.0}
net.train.run(feed)
eng.multi_stream = True     # separate launches (they honour MPNN_CONV_DBG)
for o in eng.time_ops('tr', n, reps=10):
    if o[2]:
        print('op\t%s\t%s\t%.2f' % (o[0], o[1], o[3] * 1e3))
