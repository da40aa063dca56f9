#!/usr/bin/env python3
"""The conv bodies WHERE WORK IS PLENTIFUL, launched eagerly (rocprofv3's counter collection does not survive
hipGraph replay on this stack) so that a --pmc pass can see them:

  cotrain   the joint training step of K = 8 co-trained ac_chain nets, batch 128 each (1 024 images per launch)
  eval      dense evaluation at 4 096 images
  train     one net's training step at batch 1 024

    python tools/sq_saturated.py [cotrain|eval|train ...] [REPS=4]

`tools/collect_sq_saturated.sh` runs it under rocprofv3 (kernel trace, then two SQ counter passes) and
`tools/summarize_sq_saturated.py` joins the passes into a per-body table -> profiles/r06_sq_saturated.txt."""
import os, sys
os.environ.setdefault('MPNN_GRAPH', '0')          # eager launches everywhere (lib/_plan.py: Engine.use_graph)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd')]
import torch, arch_and_hypers as A
from lib._co import CoTrainer

what = [a for a in sys.argv[1:] if not a.isdigit()] or ['cotrain', 'eval']
REPS = int(os.environ.get('REPS', '4'))
st = torch.cuda.current_stream()
g = torch.Generator().manual_seed(0)

if 'cotrain' in what:
    K, n = int(os.environ.get('CO_K', '8')), 128          # (CO_K=16: two experiments' worth of nets per launch)
    nets, feeds = [], []
    for i in range(K):
        net = A.ac_chain(k_cpt=A.k_cpts[i % 8], seed=1234 + i)((32, 32, 3), (10,))
        eng = net.engine()
        eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g)); eng.y[:n].zero_(); eng.y[:n, i % 10] = 1
        nets.append(net)
        feeds.append({net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0})
    co = CoTrainer(nets)
    for _ in range(REPS + 2): co.run(feeds)
    torch.cuda.synchronize()
    print('cotrain: %d eager joint steps of %d launches' % (REPS + 2, len(co._program(n)['ops'])), flush=True)
    del co, nets, feeds

if 'eval' in what:
    nb = int(os.environ.get('EVAL_BATCH', '4096'))
    net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
    eng = net.engine()
    eng._ensure_capacity(nb, train=False)
    eng.x0[:nb].copy_(torch.rand((nb, 32, 32, 3), generator=g)); eng.y[:nb].zero_(); eng.y[:nb, 0] = 1
    feed = {net.x0: eng.x0[:nb], net.y: eng.y[:nb]}
    for _ in range(REPS + 2): net.eval(feed)
    torch.cuda.synchronize()
    print('eval: %d eager dense passes at %d images, %d launches' % (REPS + 2, nb, len(eng.program('ev', nb, routed=False)['fwd'])), flush=True)
    del net, eng

if 'train' in what:
    n = int(os.environ.get('TRAIN_BATCH', '1024'))
    net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
    eng = net.engine()
    eng._ensure_capacity(n)
    eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g)); eng.y[:n].zero_(); eng.y[:n, 0] = 1
    feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0}
    for _ in range(REPS + 2): net.train.run(feed)
    torch.cuda.synchronize()
    print('train: %d eager steps at batch %d' % (REPS + 2, n), flush=True)
