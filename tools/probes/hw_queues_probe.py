#!/usr/bin/env python3
"""How CoGroups.plan splits 8 nets when the runtime has fewer hardware queues than groups (GPU_MAX_HW_QUEUES=1 ... 8): the
stream calibration (lib/_co.py: concurrent_streams) finds how many streams really run side by side, the split follows.
    for q in 1 2 4 8; do GPU_MAX_HW_QUEUES=$q python tools/probes/hw_queues_probe.py; done"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd')]
import torch, arch_and_hypers as A
from lib._co import CoGroups
nets = [A.ac_chain(k_cpt=k)((32, 32, 3), (10,)) for k in A.k_cpts]
cg = CoGroups.plan(nets, streams=4)
print('GPU_MAX_HW_QUEUES=%s: groups %s on %d streams, share %d' % (os.environ.get('GPU_MAX_HW_QUEUES'), [c.K for c in cg.groups], len(cg.streams), cg.share))
