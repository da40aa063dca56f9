#!/bin/bash
# Runtime environment knobs that could change what a kernel start costs (kernarg placement, graph packet path).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
TAG=base python tools/quick_step.py 400
TAG=HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=1 python tools/quick_step.py 400
TAG=HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=0 python tools/quick_step.py 400
TAG=DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/quick_step.py 400
TAG=DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python tools/quick_step.py 400
TAG=DEBUG_HIP_GRAPH_DOT_PRINT_off_GPU_MAX_HW_QUEUES=1 GPU_MAX_HW_QUEUES=1 python tools/quick_step.py 400
TAG=HSA_ENABLE_INTERRUPT=0 HSA_ENABLE_INTERRUPT=0 python tools/quick_step.py 400
TAG=AMD_SERIALIZE_KERNEL=0_HIP_LAUNCH_BLOCKING=0 AMD_SERIALIZE_KERNEL=0 HIP_LAUNCH_BLOCKING=0 python tools/quick_step.py 400
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/env_sweep.txt
