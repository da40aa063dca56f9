"""One configuration of tools/dp_corunner_probe.py for a rocprofv3 kernel trace (where do the gaps sit?).
    rocprofv3 --kernel-trace ... -- python3 tools/dp_corunner_trace.py <bucket_opt 0|1> <k> <T us> [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import dp_corunner_probe as probe
bo, k, T = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
net, eng, feed = probe.build(int(os.environ.get('RESERVE', '0')), bucket_opt=bo)
probe.install_corunner(eng, k, T, 0.0)
for _ in range(steps):
    net.train.run(feed)
torch.cuda.synchronize()
print('done')
