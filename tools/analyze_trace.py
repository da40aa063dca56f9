"""Summarise a rocprofv3 kernel-trace CSV: per-step wall time, summed kernel time, overlap."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]) for r in rows))
# take the last ~30% of the run (steady state): find repeating pattern by talr kernel
idx = [i for i, k in enumerate(ks) if 'talr_momentum' in k[2]]
if len(idx) > 6:
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
