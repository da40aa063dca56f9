#!/bin/bash
cd "$(dirname "$0")/.."
{
TAG=base python tools/quick_step.py 300
for d in 2.5 3 4; do TAG=wgdiv_$d MPNN_WG_DIV=$d python tools/quick_step.py 300; done
for w in 0.75 0.5 1.5; do TAG=latw1_$w MPNN_LAT_W1=$w python tools/quick_step.py 300; done
TAG=latw1_0.75_div3 MPNN_LAT_W1=0.75 MPNN_WG_DIV=3 python tools/quick_step.py 300
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/knob_sweep.txt
