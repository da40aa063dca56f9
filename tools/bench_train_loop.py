"""End-to-end training-loop rate INCLUDING the input pipeline (host RNG draws + device augmentation),
as multipath-nn_amd/train-nets runs it; bench.py's headline number has the batch resident in HBM."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import numpy as np, torch
import arch_and_hypers as A
from lib.data import Dataset, _draw_augmentation, _sym_of_sources

ds = Dataset.synthetic(n_tr=4096)
net = A.ac_chain(k_cpt=0.0)((32, 32, 3), (10,))
eng = net.engine(); eng._ensure_capacity(128); ds.to_device('cuda:0')
np.random.seed(0)


def step(t):
    x0, y = ds.augmented_training_batch_device(128, x_out=eng.x0[:128], y_out=eng.y[:128])
    net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: A.λ_lrn(t), net.τ: A.τ_ds(t)})


for t in range(20): step(t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(500): step(t)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
print('training loop with input pipeline: %.3f ms/step = %.0f img/s' % (dt * 1e3, 128 / dt))
sym = _sym_of_sources(ds.y_tr, ds.m_sym)
t0 = time.perf_counter()
for _ in range(200): _draw_augmentation(128, len(ds.x0_tr), ds.y_tr, ds.m_sym, 4, sym)
print('host RNG draws alone: %.3f ms/batch' % ((time.perf_counter() - t0) / 200 * 1e3))
