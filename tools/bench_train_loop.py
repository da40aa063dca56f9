"""End-to-end training-loop rate INCLUDING the input pipeline (host RNG draws + upload + device augmentation), as
multipath-nn_amd/train-nets runs it, beside the step with the batch resident in HBM (what bench.py times).

    python tools/bench_train_loop.py [steps]     -> profiles/r04_train_loop.txt
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import numpy as np, torch
import arch_and_hypers as A
from lib.data import Dataset, _draw_augmentation, _draw_augmentation_fast, _sym_of_sources

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = 128


def make():
    ds = Dataset.synthetic(n_tr=50000)              # CIFAR-10's training-set size (614 MB resident)
    net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
    eng = net.engine(); eng._ensure_capacity(N); ds.to_device('cuda:0')
    np.random.seed(0)
    return ds, net, eng


def timed(step, label):
    for t in range(30): step(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(steps): step(t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print('%-78s %.3f ms/step = %7.0f img/s' % (label, dt * 1e3, N / dt), flush=True)
    return dt


ds, net, eng = make()
feed = lambda x0, y, t: {net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: A.λ_lrn(t), net.τ: A.τ_ds(t)}
x0, y = eng.x0[:N], eng.y[:N]
x0.uniform_(); y.zero_(); y[:, 0] = 1
base = timed(lambda t: net.train.run(feed(x0, y, t)), 'batch resident in HBM, schedules fed per step (no input pipeline)')


def eager(t):
    a, b = ds.augmented_training_batch_device(N, x_out=eng.x0[:N], y_out=eng.y[:N])
    net.train.run(feed(a, b, t))
timed(eager, 'pipeline, augmentation launched eagerly before the step graph')

x0, y = ds.bind_engine(eng, N)


def bound(t):
    ds.stage_training_draws(N, eng=eng)
    net.train.run(feed(x0, y, t))
full = timed(bound, 'pipeline, one hipGraph replay per step (draws + async upload; augmentation inside the step graph)')
print('pipeline / resident: %.3f' % (full / base))


def bound4(t):
    # (as train-nets runs it since round 5: four iterations per hipGraph replay, Engine.run_steps; `timed` counts calls)
    ds.stage_training_draws_k(4, N, eng=eng)
    net.train.run_steps([feed(x0, y, 4 * t + j) for j in range(4)])
full4 = timed(bound4, 'pipeline as train-nets runs it: FOUR steps per replay (time per call = 4 steps)') / 4
print('  -> %.3f ms per step = %.0f img/s;  pipeline / resident: %.3f' % (full4 * 1e3, N / full4, full4 / base))

sym = _sym_of_sources(ds.y_tr, ds.m_sym)
t0 = time.perf_counter()
for _ in range(200): _draw_augmentation(N, len(ds.x0_tr), ds.y_tr, ds.m_sym, 4, sym)
print('host draws alone, per-call numpy loop (rounds 1-3): %.3f ms/batch' % ((time.perf_counter() - t0) / 200 * 1e3))
t0 = time.perf_counter()
for _ in range(2000): _draw_augmentation_fast(N, len(ds.x0_tr), ds._sym_u8, 4, all_sym=ds._all_sym)
print('host draws alone, replayed over raw words (mpnn_draw_augmentation): %.3f ms/batch' % ((time.perf_counter() - t0) / 2000 * 1e3))
