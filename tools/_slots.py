import sys, os
sys.path.insert(0, '/root/repo/multipath-nn_amd')
from lib import _hip
import torch
torch.zeros(1, device='cuda')
lib = _hip.load()
for (H, C) in ((4, 128), (4, 64), (8, 64), (4, 32), (8, 32), (16, 32), (32, 16), (4, 16)):
    print(H, C, lib.mpnn_msconv_bwd_scale_slots(H, H, C, 1))
