#!/bin/bash
# Instruction-cache counters per kernel of the training step (eager launches: counter collection does not work inside
# hipGraph replay on this stack).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/icache
rm -rf $O; mkdir -p $O
MPNN_GRAPH=0 timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES --kernel-trace --output-format csv -d $O -o ic -- python3 $R/tools/quick_step.py 60 > /dev/null 2> $O/err.txt
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/icache/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[(r['Kernel_Name'].split('(')[0][:44], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
print('%-44s %8s %5s %10s %10s %10s %8s' % ('kernel', 'grid', 'n', 'req', 'miss', 'miss_dup', 'waves'))
for k, c in sorted(acc.items()):
    o = {n: sum(v) / len(v) for n, v in c.items()}
    print('%-44s %8s %5d %10.0f %10.0f %10.0f %8.0f' % (k[0], k[1], len(c['SQC_ICACHE_REQ']), o.get('SQC_ICACHE_REQ', 0), o.get('SQC_ICACHE_MISSES', 0), o.get('SQC_ICACHE_MISSES_DUPLICATE', 0), o.get('SQ_WAVES', 0)))
PY
rm -rf $O
