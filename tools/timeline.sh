#!/bin/bash
# Step timeline (per-kernel durations inside the replayed graph) of the bench configuration under rocprofv3.
#   gpurun -- 'bash tools/timeline.sh [name]'   -> gpurun_out/timeline_<name>.txt ; environment knobs pass through
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
N=${1:-cur}
O=$R/gpurun_out/tl_$N
rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 $R/tools/quick_step.py 100 > $O/run.txt 2> $O/err.txt
cd $R
python tools/analyze_trace.py $O/kt > gpurun_out/timeline_$N.txt
rm -rf $O
cat gpurun_out/timeline_$N.txt
