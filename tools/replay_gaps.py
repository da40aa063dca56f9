"""Idle time between consecutive training steps in a rocprofv3 kernel trace (the gap between finish_opt_k and the next
step's first kernel): python tools/replay_gaps.py <trace dir>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
idx = [i for i, k in enumerate(ks) if 'finish_opt_k' in k[2]]
sel = idx[-82:-1]
gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in sel[:-1]]
walls = [(ks[sel[j + 1]][1] - ks[sel[j]][1]) / 1e3 for j in range(len(sel) - 1)]
print('last %d steps: mean step %.1f us; idle between finish_opt_k and the next step\'s first kernel (us): %s'
      % (len(walls), sum(walls) / len(walls), ' '.join('%.1f' % g for g in gaps[:16])))
print('mean idle per step %.2f us' % (sum(gaps) / len(gaps)))
