"""Per-launch determinism probe: run the forward launches eagerly, three times from the same
state, and report the first launch whose outputs differ between repetitions."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import numpy as np, torch, arch_and_hypers as A
net = A.ac_chain(k_cpt=1.6e-8, seed=5)((32, 32, 3), (10,)); eng = net.engine()
rng = np.random.default_rng(0)
x0 = rng.random((128, 32, 32, 3)).astype(np.float32); y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 128)]
feed = {net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.τ: 1.0}
for _ in range(2): net.train.run(feed)
torch.cuda.synchronize()
P0, A0, S0 = eng.P.clone(), eng.A.clone(), eng.S.clone()

def snap():
    d = {}
    for k, b in enumerate(eng.blocks):
        for i in range(b.L):
            d['s%d_%d' % (k, i)] = b.s[i].clone()
            if i < b.L - 1: d['sp%d_%d' % (k, i)] = b.sp[i].clone()
    d['dsum'] = eng.batch_stat_sums().clone()
    return d

def fwd_only(group, reps=4):
    eng.group_fwd = group
    prog = eng.program('tr', 128)
    ops = [o for o in prog['fwd'] if o.what not in ('fork', 'join')]
    st = torch.cuda.current_stream()
    out = []
    for rep in range(reps):
        eng.P.copy_(P0); eng.A.copy_(A0); eng.S.copy_(S0); eng.invalidate_packs()
        eng._zero(True); eng._pack()
        per = []
        for op in ops:
            op(st.cuda_stream)
            if op.what in ('fwd_group', 'msconv_fwd'):
                torch.cuda.synchronize(); per.append((op.what + ' ' + op.tag, snap()))
        out.append(per)
    return out

res = {}
for group in (True, False):
    out = fwd_only(group)
    res[group] = out
    print('group_fwd =', group)
    for j, (name, ref) in enumerate(out[0]):
        bad = []
        for rep in range(1, len(out)):
            for k in ref:
                e = (out[rep][j][1][k].double() - ref[k].double()).abs().max().item()
                if e > 0: bad.append((rep, k, '%.2e' % e))
        print('  launch %2d %-60s %s' % (j, name[:60], bad[:6] if bad else 'ok'))
# grouped vs single launches (final state)
a, b = res[True][0][-1][1], res[False][0][-1][1]
print('grouped vs single:', [(k, '%.2e' % (a[k].double() - b[k].double()).abs().max().item()) for k in a if k != 'dsum' and (a[k] != b[k]).any()][:12])
