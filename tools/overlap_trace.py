"""Concurrency of a rocprofv3 kernel trace: how many kernels are in flight over the steady-state part of a run.
    python tools/overlap_trace.py <rocprof output dir> [fraction of the run to look at, from the end: 0.5]
For the side-by-side form of co-training (lib/_co.py: CoGroups) -- per hardware queue: kernels, busy time; overall: the
fraction of the wall time with 0 / 1 / 2 / ... kernels running, mean kernels in flight, mean duration per kernel name."""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48], r.get('Queue_Id', '?')) for r in rows)
t_end = max(k[1] for k in ks)
t0 = t_end - int((t_end - ks[0][0]) * frac)
ks = [k for k in ks if k[0] >= t0 and 'spin_k' not in k[2] and 'noop_k' not in k[2]]
wall = (max(k[1] for k in ks) - ks[0][0]) / 1e3
print('window: %.1f ms, %d kernels' % (wall / 1e3, len(ks)))
by_q = collections.defaultdict(lambda: [0, 0.0])
for s, e, _, q in ks:
    by_q[q][0] += 1; by_q[q][1] += (e - s) / 1e3
for q, (c, b) in sorted(by_q.items()):
    print('  queue %-4s %6d kernels, busy %5.1f %% of the window' % (q, c, 100 * b / wall))
ev = sorted([(s, 1) for s, _, _, _ in ks] + [(e, -1) for _, e, _, _ in ks])
depth, prev, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[depth] += t - prev
    depth += d; prev = t
tot = sum(hist.values())
print('kernels in flight: ' + '  '.join('%d: %.1f %%' % (k, 100 * v / tot) for k, v in sorted(hist.items())))
print('mean kernels in flight %.2f' % (sum(k * v for k, v in hist.items()) / tot))
dur = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, _ in ks:
    dur[n][0] += 1; dur[n][1] += (e - s) / 1e3
print('%-50s %7s %9s %9s' % ('kernel', 'calls', 'mean us', 'total ms'))
for n, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:14]:
    print('%-50s %7d %9.1f %9.2f' % (n, c, t / c, t / 1e3))
