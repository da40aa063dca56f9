"""Per-workgroup phase timeline of every conv / backward launch of one training step.

Needs the trace build of the library (make -C multipath-nn_amd/csrc trace) and selects it through
MPNN_HIP_LIB.  For each launch: when workgroups start (dispatch ramp), and when they pass the phase
stamps (tables ready, first tile staged, first unit done, loop done, exit), relative to the first
workgroup's start; clock = 100 MHz (10 ns ticks).

    python tools/trace_phases.py [substring of the launch tag]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('MPNN_HIP_LIB', os.path.join(ROOT, 'multipath-nn_amd', 'libmpnn_hip_trace.so'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import numpy as np, torch, arch_and_hypers as A
from lib import _hip

want = sys.argv[1] if len(sys.argv) > 1 else ''
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
n = int(os.environ.get('BATCH', '128'))
eng._ensure_capacity(n)
eng.x0[:n].uniform_(); eng.y[:n].zero_(); eng.y[:n, 0] = 1
eng.co_share = int(os.environ.get('CO_SHARE', '1'))      # (the planner setting of a co-trained group: grids for slots / share, level launches)
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0}
for _ in range(3): net.train.run(feed)
torch.cuda.synchronize()
st = torch.cuda.current_stream()
CO_K = int(os.environ.get('CO_K', '0'))
if CO_K > 1:
    # the MERGED launches of CO_K co-trained nets (lib/_co.py: one launch per layer for all of them): what the workgroups of
    # a saturated launch do, all nets' workgroups contending.  Rows are grouped by (member, body) over all nets.
    from lib._co import CoTrainer
    nets, feeds = [net], [feed]
    for i in range(1, CO_K):
        nt = A.ac_chain(k_cpt=A.k_cpts[i % 8], seed=1234 + i)((32, 32, 3), (10,))
        e = nt.engine()
        e.x0[:n].uniform_(); e.y[:n].zero_(); e.y[:n, i % 10] = 1
        nets.append(nt)
        feeds.append({nt.x0: e.x0[:n], nt.y: e.y[:n], nt.mode: 'tr', nt.λ_lrn: 0.1, nt.τ: 1.0})
    eng.co_share = 1
    co = CoTrainer(nets)
    for _ in range(3): co.run(feeds)
    torch.cuda.synchronize()
    co.use_graph = False
    for e in co.engs: e.mark_dirty()
    for e in co.engs: e._begin(True)
    ops = list(co._program(n)['ops'])
else:
    prog = eng.program('tr', n)
    ops = [o for o in list(prog['fwd']) + list(prog['bwd']) if o.what not in ('fork', 'join')]
    eng._zero(True); eng._pack()
for op in ops: op(st.cuda_stream)
torch.cuda.synchronize()
NWG = 1 << 16
SL = 12
buf = torch.zeros(NWG * SL, dtype=torch.int64, device=eng.dev)
KIND = {1: 'fwd', 2: 'dgh_bn', 3: 'dgh_raw', 4: 'dgv', 8: 'wgrad', 10: 'lin_fwd', 11: 'lin_bwd', 12: 'tail_fwd', 13: 'tail_bwd', 14: 'route'}
NAMES = ['start', 'tables', 'staged', 'unit0', 'loop', 'exit']
for op in ops:
    if op.what not in ('fwd_group', 'msconv_fwd', 'bwd_scale', 'bwd_level', 'lin_fwd', 'lin_bwd', 'exit_tail_fwd', 'exit_tail_bwd', 'route') or (want not in op.tag and want != op.what):
        continue
    for _ in range(2): op(st.cuda_stream)            # warm caches
    torch.cuda.synchronize()
    buf.zero_(); torch.cuda.synchronize()
    _hip.check(eng.lib.mpnn_debug_set_trace(buf.data_ptr()), 'set_trace')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st); op(st.cuda_stream); e1.record(st); e1.synchronize()
    _hip.check(eng.lib.mpnn_debug_set_trace(None), 'set_trace')
    t = buf.cpu().numpy().reshape(NWG, SL)
    if os.environ.get('TRACE_DUMP'):              # raw rows (index = blockIdx.x) of every traced launch, for offline analysis
        os.makedirs(os.environ['TRACE_DUMP'], exist_ok=True)
        nz = np.nonzero(t[:, 0])[0]
        np.save(os.path.join(os.environ['TRACE_DUMP'], '%02d_%s.npy' % (ops.index(op), op.what)), np.concatenate([nz[:, None], t[nz]], axis=1))
    rows = t[t[:, 0] != 0]
    if not len(rows):
        continue
    t0 = rows[:, 0].min()
    span = (rows[:, 5].max() - t0) / 100.0
    ev_t = sorted([(int(r[0]), 1) for r in rows if r[5]] + [(int(r[5]), -1) for r in rows if r[5]])
    cur = peak = 0
    for _, dlt in ev_t:
        cur += dlt; peak = max(peak, cur)
    print('\n%s [%s]  event %.1f us, first start -> last exit %.1f us, %d workgroups, peak concurrency %d' % (op.what, op.tag, e0.elapsed_time(e1) * 1e3, span, len(rows), peak))
    for mem, kind in sorted(set(zip(rows[:, 11], rows[:, 6]))):
        r = rows[(rows[:, 6] == kind) & (rows[:, 11] == mem)]
        if mem:
            print(' member %d:' % (mem - 1))
        rel = (r[:, :6] - t0) / 100.0
        rel2 = (r[:, 8:11] - t0) / 100.0
        units = r[:, 7]
        print('  %-7s %4d wgs, units/wg %d..%d' % (KIND.get(int(kind), str(kind)), len(r), units.min(), units.max()))
        for k in range(6):
            col = rel[:, k][r[:, k] != 0]
            if len(col):
                print('     %-7s min %6.2f  med %6.2f  max %6.2f us' % (NAMES[k], col.min(), np.median(col), col.max()))
        for k, nm in enumerate(('u0 mfma', 'u0 stage', 'u0 epi')):
            col = rel2[:, k][r[:, 8 + k] != 0]
            if len(col):
                print('     %-8s min %6.2f  med %6.2f  max %6.2f us' % (nm, col.min(), np.median(col), col.max()))
        ok = (units > 1) & (r[:, 3] != 0) & (r[:, 4] != 0)
        if ok.any():
            pu = (r[ok, 4] - r[ok, 3]) / 100.0 / (units[ok] - 1)
            print('     steady state: %.3f us per unit (median; min %.3f max %.3f)' % (np.median(pu), pu.min(), pu.max()))
        d = (r[:, 5] - r[:, 0]) / 100.0
        print('     in-wg time (start->exit): min %.2f med %.2f max %.2f us' % (d.min(), np.median(d), d.max()))
