#!/bin/bash
# Budget knobs of the backward launches (weight-gradient share of the resident slots, relative item latencies of
# the level budget): re-run after changes to the bodies.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
TAG=base python tools/quick_step.py 300
for d in 1.6 2.5 3; do TAG=wgdiv_$d MPNN_WG_DIV=$d python tools/quick_step.py 300; done
for w in 0.75 1.3; do TAG=latw1_$w MPNN_LAT_W1=$w python tools/quick_step.py 300; done
TAG=nolevels MPNN_BWD_LEVELS=0 python tools/quick_step.py 300
TAG=noxcd MPNN_XCD=0 python tools/quick_step.py 300
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/knob_sweep.txt
