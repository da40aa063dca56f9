#!/bin/bash
# Kernel timeline of the one-graph data-parallel step with the co-runner stand-in.
#   gpurun -- 'bash tools/dp_trace.sh <name> <bucket_opt> <k> <T>'  -> gpurun_out/dp_trace_<name>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
N=${1:-cur}
O=$R/gpurun_out/dptl_$N
rm -rf "$O"; mkdir -p "$O"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 $R/tools/dp_corunner_trace.py $2 $3 $4 > $O/run.txt 2> $O/err.txt
cd $R
python tools/analyze_trace.py $O/kt > gpurun_out/dp_trace_$N.txt
rm -rf "$O"
