#!/bin/bash
# Per-kernel A/B of two builds of the library: rocprofv3 kernel stats of tools/quick_step.py with the default library and
# with MPNN_HIP_LIB=$1, average duration per kernel side by side.     bash tools/ab_kernel_diff.sh <path to the variant .so>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/ab; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -o kt -- python3 $R/tools/quick_step.py 300 > /dev/null 2> $O/a.err
MPNN_HIP_LIB=$1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -o kt -- python3 $R/tools/quick_step.py 300 > /dev/null 2> $O/b.err
python3 - $O <<'PY'
import csv, glob, sys
def load(d):
    f = sorted(glob.glob(d + '/**/*kernel_stats.csv', recursive=True))[-1]
    return {r['Name'].split('(')[0][:60]: (int(r['Calls']), float(r['AverageNs']) / 1e3) for r in csv.DictReader(open(f))}
a, b = load(sys.argv[1] + '/a'), load(sys.argv[1] + '/b')
tot_a = tot_b = 0.0
print('%-62s %6s %9s %9s %8s' % ('kernel', 'calls', 'base us', 'variant', 'delta'))
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1]):
    if k in b and a[k][0] > 100:
        per_step = a[k][0] / max(v[0] for v in a.values() if True) 
        print('%-62s %6d %9.2f %9.2f %+8.2f' % (k, a[k][0], a[k][1], b[k][1], b[k][1] - a[k][1]))
        tot_a += a[k][0] * a[k][1]; tot_b += b[k][0] * b[k][1]
print('sum of kernel time: base %.1f ms, variant %.1f ms (%+.2f %%)' % (tot_a / 1e3, tot_b / 1e3, 100 * (tot_b - tot_a) / tot_a))
PY
rm -rf $O/a $O/b
