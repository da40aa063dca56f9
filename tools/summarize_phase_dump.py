"""Balance of the workgroups inside every conv launch, from the raw rows of the in-kernel phase trace
(CO_K=8 TRACE_DUMP=<dir> python tools/trace_phases.py > <txt>; then python tools/summarize_phase_dump.py <txt> <dir>).

Per launch: event time, first start -> last exit, workgroups, and MEAN WORKGROUP BUSY TIME / SPAN -- the share of the resident
slots' time in which their workgroup is still running (1 - that = slots waiting for the launch's slowest workgroup);
per (member, body): workgroups, units per workgroup (min..max), when its workgroups end (median, max)."""
import glob, re, sys
import numpy as np

KIND = {1: 'fwd', 2: 'dgh_bn', 3: 'dgh_raw', 4: 'dgv', 8: 'wgrad'}
txt, dump = sys.argv[1], sys.argv[2]
tags = []
for ln in open(txt):
    m = re.match(r'^(\w+) \[(.*?)\]\s+event ([\d.]+) us, first start -> last exit ([\d.]+)', ln)
    if m:
        tags.append((m.group(1), m.group(2), float(m.group(3))))
files = [f for f in sorted(glob.glob(dump + '/*.npy')) if len(np.load(f))]
assert len(files) == len(tags), (len(files), len(tags))
tot_ev = tot_idle = 0.0
for f, (what, tag, ev) in zip(files, tags):
    assert what in f, (f, what)
    if what not in ('fwd_group', 'bwd_scale'):
        continue
    t = np.load(f)[:, 1:]
    t0 = t[:, 0].min()
    end, start = (t[:, 5] - t0) / 100.0, (t[:, 0] - t0) / 100.0
    span = end.max()
    busy = (end - start).sum() / (len(t) * span)
    print('%-10s %-46s event %6.1f span %6.1f wgs %4d  mean wg busy / span = %.2f' % (what, tag[:46], ev, span, len(t), busy))
    for mem, kind in sorted(set(zip(t[:, 11], t[:, 6]))):
        r = (t[:, 6] == kind) & (t[:, 11] == mem)
        u = t[r, 7]
        print('      m%d %-7s wgs %4d units %3d..%-3d  end med %6.1f max %6.1f'
              % (mem - 1 if mem else 0, KIND.get(int(kind), kind), r.sum(), u.min(), u.max(), np.median(end[r]), end[r].max()))
    tot_ev += ev
    tot_idle += ev * (1 - busy)
print('sum of the conv launches\' events %.0f us; slots waiting for the slowest workgroup, event-weighted: %.0f us (%.0f %%)'
      % (tot_ev, tot_idle, 100 * tot_idle / tot_ev))
