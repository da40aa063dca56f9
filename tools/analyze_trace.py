"""Summarise a rocprofv3 kernel-trace CSV: per-step wall time, summed kernel time, overlap."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]) for r in rows))
# take the last ~30% of the run (steady state): find the repeating pattern by the launch that ends a training step
# (the optimizer; since round 4 the fused finish_opt_k in single-process runs)
last = 'finish_opt_k' if sum('finish_opt_k' in k[2] for k in ks) > 6 else 'talr_momentum'
idx = [i for i, k in enumerate(ks) if last in k[2]]
if len(idx) > 24:
    # the last 20 complete steps: per-step wall, busy and idle time; the MEDIAN step is printed kernel by kernel (a
    # single step now and then carries a profiler hiccup of tens of microseconds)
    steps_ = []
    for j in range(len(idx) - 21, len(idx) - 1):
        seg = ks[idx[j] + 1:idx[j + 1] + 1]
        prev_end = ks[idx[j]][1]
        wall = (seg[-1][1] - prev_end) / 1e3
        busy = sum(e - s for s, e, _ in seg) / 1e3
        steps_.append((wall, busy, seg, prev_end))
    walls = sorted(w for w, _, _, _ in steps_)
    med = walls[len(walls) // 2]
    wall, busy, one, prev_end = min(steps_, key=lambda t: abs(t[0] - med))
    print('per step (median of the last 20): wall %.1f us (min %.1f, max %.1f), sum of kernel durations %.1f us, idle %.1f us, kernels %d'
          % (med, walls[0], walls[-1], busy, wall - busy, len(one)))
    print('%-58s %8s %8s' % ('kernel (the median step, in order)', 'dur us', 'gap us'))
    prev_e = prev_end
    for s, e, name in one:
        print('%-58s %8.1f %8.1f' % (name[:58], (e - s) / 1e3, (s - prev_e) / 1e3))
        prev_e = e
elif len(idx) > 6:
    a, b = idx[-4], idx[-1]
    seg = ks[a + 1:b + 1]
    steps = 3
    wall = (seg[-1][1] - seg[0][0]) / steps / 1e3
    busy = sum(e - s for s, e, _ in seg) / steps / 1e3
    # union of intervals
    cur_s, cur_e, union = seg[0][0], seg[0][1], 0
    for s, e, _ in seg[1:]:
        if s > cur_e:
            union += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print('per step: wall %.1f us, sum of kernel durations %.1f us, GPU busy (union) %.1f us, idle %.1f us, kernels %d'
          % (wall, busy, union / steps / 1e3, wall - union / steps / 1e3, len(seg) // steps))
    # per-kernel mean duration and mean gap to the previous kernel's end, one steady-state step
    one = seg[-(len(seg) // steps):]
    print('%-58s %8s %8s' % ('kernel (last step, in order)', 'dur us', 'gap us'))
    prev_e = None
    for s, e, name in one:
        print('%-58s %8.1f %8.1f' % (name[:58], (e - s) / 1e3, 0.0 if prev_e is None else (s - prev_e) / 1e3))
        prev_e = e
