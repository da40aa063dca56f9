"""How much would the backward launches of ONE dependency level overlap if they ran at the same time?
For each wavefront level of the backward pass (B(b, S) needs B(b+1, S) and B(b, S+1)) the member launches are
timed (a) back to back on one stream and (b) each on its own stream, all streams saturated (R repeats per
stream): an upper bound of what a fused launch of the level can reach, each member at its own occupancy."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import numpy as np, torch, arch_and_hypers as A
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
n = 128
g = torch.Generator().manual_seed(0)
eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g)); eng.y[:n].zero_(); eng.y[:n, 0] = 1
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0}
eng.use_graph = False
for _ in range(3): net.train.run(feed)
torch.cuda.synchronize()
prog = eng.program('tr', n)
ops = [o for o in prog['bwd'] if o.what == 'bwd_scale']
# order of the program: blocks reversed, scales coarsest first
keys = []
for kb, b in enumerate(reversed(eng.blocks)):
    bi = len(eng.blocks) - 1 - kb
    for i in range(b.L - 1, -1, -1):
        S = 4 - b.L + i                      # absolute scale index: 0 = 32 px ... 3 = 4 px
        keys.append((bi, S))
assert len(keys) == len(ops)
levels = {}
for k, (bi, S) in enumerate(keys):
    levels.setdefault((7 - bi) + (3 - S), []).append(k)
R = 50
streams = [torch.cuda.Stream() for _ in range(4)]
main = torch.cuda.current_stream()

def t_serial(idx):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in idx: ops[k](main.cuda_stream)
    e0.record(main)
    for _ in range(R):
        for k in idx: ops[k](main.cuda_stream)
    e1.record(main); e1.synchronize()
    return e0.elapsed_time(e1) / R * 1e3

def t_conc(idx):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(main)
    ends = []
    for s, k in zip(streams, idx):
        s.wait_event(e0)
    for r in range(R):
        for s, k in zip(streams, idx):
            ops[k](s.cuda_stream)
    for s, k in zip(streams, idx):
        e = torch.cuda.Event(enable_timing=True); e.record(s); ends.append(e)
    for e in ends: e.synchronize()
    return max(e0.elapsed_time(e) for e in ends) / R * 1e3

tot_s = tot_c = tot_m = 0.0
for d in sorted(levels):
    idx = levels[d]
    each = [t_serial([k]) for k in idx]
    ser = t_serial(idx)
    con = t_conc(idx) if len(idx) > 1 else ser
    tot_s += ser; tot_c += con; tot_m += max(each)
    print('level %2d: %-60s each %s  serial %6.1f  concurrent %6.1f  (max member %5.1f)' % (
        d, ' | '.join('%s[b%d]' % (ops[k].tag, keys[k][0]) for k in idx), ' '.join('%5.1f' % t for t in each), ser, con, max(each)))
print('sum: serial %.1f us, concurrent (saturated streams) %.1f us, sum of max members %.1f us' % (tot_s, tot_c, tot_m))
eng.mark_dirty()          # (program ops were launched outside run(): the next step must clear the accumulators)
