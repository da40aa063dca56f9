#!/bin/bash
# Does any runtime knob of the hipGraph executor lower the cost of a parallel branch inside the step graph?
#   gpurun -- 'bash tools/dp_env_sweep.sh'  -> gpurun_out/r04_dp_env_sweep.txt
mkdir -p gpurun_out
O=gpurun_out/r04_dp_env_sweep.txt
: > $O
run() { TAG="$1" env $1 python tools/dp_corunner_probe.py --one 0 0 16 40 >> $O 2>/dev/null; }
python tools/dp_corunner_probe.py --one 0 0 0 0 >> $O 2>/dev/null
run "BASE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=1"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=2"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=4"
run "DEBUG_HIP_GRAPH_BATCH_SIZE=1"
run "DEBUG_HIP_GRAPH_BATCH_SIZE=64"
run "GPU_MAX_HW_QUEUES=8"
run "HIP_FORCE_DEV_KERNARG=0"
run "MPNN_GRAPH=0"
run "MPNN_DP_ONE_GRAPH=0"
cat $O
