#!/usr/bin/env python3
"""Headline benchmark: images/sec of full training steps (forward + backward +
gradient all-reduce + TALR/momentum update) of the CIFAR-10 actor-routed chain
net (BASELINE.json: cifar10-ac), batch 128 per GPU, synthetic 32x32x3 inputs
resident in HBM, fp32 (the reference's arithmetic type).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract: see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'multipath-nn_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

# (before anything initialises HIP: the host driver of this pool only supports dmabuf IPC -- without it RCCL between
# processes fails with hipIpcGetMemHandle: invalid argument; the launcher below sets it for its children as well)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np
import torch

MAC_FWD = 20699872                    # ac/cr chain forward MAC per image
F_TRAIN = 6 * MAC_FWD - 2 * 587520    # no dgrad w.r.t. the image: 123.024 MFLOP
PEAK_F32_MFMA = 157.3                 # TFLOP/s, MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32)


def synthetic(n, seed, device):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.rand((n, 32, 32, 3), generator=g)
    y = torch.nn.functional.one_hot(torch.randint(0, 10, (n,), generator=g), 10).float()
    return x0.to(device), y.to(device)


def csrc_sha16():
    """Hash of the kernel sources and the launch plan: a committed PMC summary is only quoted while it
    still describes the code that runs (tools/summarize_pmc.py stores the same hash)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'multipath-nn_amd', 'csrc', '*.h*'))) + \
        [os.path.join(ROOT, 'multipath-nn_amd', 'lib', '_plan.py')] + sorted(glob.glob(os.path.join(ROOT, 'multipath-nn_amd', 'lib', '_eng_*.py')))
    for f in files:
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def set_exit_fractions(net, feed, n, fractions):
    """Shift the exit-sink bias of every router so that fractions[k] of the batch leaves at exit k
    (chain nets: sink 0 = the exit's classifier, sink 1 = the next block)."""
    net.eval(feed)
    alive = np.ones(n, bool)
    for k, ℓ in enumerate(net.switches):
        r = ℓ.router.x.cpu().numpy().astype(np.float64)
        want = int(round(fractions[k] * n))
        margin = r[:, 1] - r[:, 0]
        idx = np.flatnonzero(alive)
        order = idx[np.argsort(margin[idx], kind='stable')]
        if want <= 0 or len(idx) == 0:
            shift = (margin[idx].min() - 1.0) if len(idx) else 0.0
        elif want >= len(idx):
            shift = margin[idx].max() + 1.0
        else:
            shift = 0.5 * (margin[order[want - 1]] + margin[order[want]])
        b = ℓ.router.comps[-1].params.b
        v = b.numpy().copy()
        v[0] += shift
        b.assign(v)
        alive[order[:max(want, 0)]] = False


CLOCK_WARMUP = 96        # untimed training steps in front of the warm-up steps: the GPU's clocks after the idle time of graph capture (main())


def cpu_baseline(batch, seconds=24.0):
    """The oracle (oracle/ref_net.py, torch-CPU fp32, all host cores) on the same step."""
    import arch_and_hypers as A
    from oracle.ref_net import RefNet
    net = A.ac_chain(k_cpt=0.0)((32, 32, 3), (10,))
    rng = np.random.default_rng(1234)
    vals = {}
    for p in net._all_params:
        kind, scale = p.init
        vals[id(p)] = ((scale * rng.standard_normal(p.size)) if kind == 'normal' else
                       (np.ones(p.size) if kind == 'ones' else np.zeros(p.size))).reshape(p.shape)
    ref = RefNet(net, torch.float32)
    ref.load_params(vals)
    x0, y = synthetic(batch, 0, 'cpu')
    x0, y = x0.numpy(), y.numpy()
    all_cores = torch.get_num_threads()
    best = None
    # torch-CPU oversubscribes on many-core hosts for this small problem: time it with all host
    # threads and with 32 / 16, and report the fastest (cores = threads actually used).
    tries = sorted({all_cores, min(all_cores, 32), min(all_cores, 16)}, reverse=True)
    for threads in tries:
        torch.set_num_threads(threads)
        times = []
        t_end = time.time() + seconds / len(tries)
        for i in range(1000):
            t0 = time.time()
            ref.train_step(x0, y, 0.1, τ=1.0)
            if i >= 1:
                times.append(time.time() - t0)
            if time.time() > t_end and len(times) >= 3:
                break
        med = float(np.median(times))
        if best is None or med < best[0]:
            best = (med, threads, len(times))
    torch.set_num_threads(all_cores)
    med, threads, cnt = best
    return dict(value=batch / med, unit='images/s', cores=threads, kind='port',
                sample='%d timed steps of batch %d (median %.1f ms/step) with %d of %d host threads (fastest of %s), '
                       'oracle/ref_net.py torch-CPU fp32' % (cnt, batch, med * 1e3, threads, all_cores, tries))


def spawn_ranks(n_ranks):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_ranks),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes needs it on this stack
    env.setdefault('OMP_NUM_THREADS', '4')
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def time_replays(run, reps, chunk=10):
    """Median ms per call of `run` over reps calls: HIP events on the launch stream around chunks."""
    st = torch.cuda.current_stream()
    k = max(1, reps // chunk)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    evs[0].record(st)
    for i in range(k):
        for _ in range(chunk):
            run()
        evs[i + 1].record(st)
    torch.cuda.synchronize()
    per = np.array([evs[i].elapsed_time(evs[i + 1]) / chunk for i in range(k)])
    return float(np.median(per))


def measure_dp_structure(net, eng, feed, single_ms):
    """What the data-parallel STRUCTURE costs before any wire time: the step as the multi-GPU run executes
    it -- the gradient-bucket sections with an asynchronous RCCL all-reduce per bucket on the process group's
    stream, the waits and the optimizer, captured as one hipGraph (or, where collectives do not capture, one
    graph per section) -- with a ONE-rank RCCL process group, beside the single-graph step of the headline."""
    import torch.distributed as dist
    from lib import _dp
    _dp.init(backend='nccl', force=True)
    try:
        _dp.attach(net, force=True)
        for _ in range(5):
            net.train.run(feed)
        torch.cuda.synchronize()
        ms1 = time_replays(lambda: net.train.run(feed), 200)
        whole = eng.step_graph_form() == 'whole'
        # the form the multi-GPU run takes: K steps -- K all-reduces -- per hipGraph replay, as the single-process headline
        K = 4
        for _ in range(3):
            net.train.run_steps([feed] * K)
        torch.cuda.synchronize()
        k_captured = eng.k_step_graph_captured(dp=True)
        ms = time_replays(lambda: net.train.run_steps([feed] * K), 50, chunk=5) / K if k_captured else ms1
        return {'ms_per_step': ms, 'ms_per_step_single_graph': single_ms, 'ratio': ms / single_ms,
                'steps_per_graph': K if k_captured else 1, 'ms_per_step_one_step_graphs': ms1, 'ratio_one_step_graphs': ms1 / single_ms,
                'buckets': list(eng.dp_buckets), 'collectives_per_step': len(eng.dp_buckets), 'rccl_ranks': dist.get_world_size(),
                'reserved_cus': eng.dp_reserve_cus, 'per_bucket_update': eng.per_bucket_update,
                'form': 'ONE hipGraph per step: bucket sections + RCCL all-reduces (captured on the process group\'s stream) + optimizer'
                        if whole else 'one hipGraph per bucket section, all-reduces issued from the host, optimizer graph',
                'what': 'forced 1-rank RCCL group, 200 steps, beside the single-process one-graph step'}
    finally:
        _dp.detach(net)
        dist.destroy_process_group()


def measure_corunner(single_ms, n, k=16, T=40.0):
    """Robustness of the data-parallel step to a kernel that runs BESIDE it (tools/dp_corunner_probe.py), on one GPU:
    each bucket's collective replaced by k workgroups x 512 threads that hold their slots for T us, inside the one-graph
    step.  `one_bucket`: the shipped form (the stand-in sits between the end of the backward pass and the optimizer, so
    T is exposed by construction; `minus_T` is what the parallel branch itself costs).  `three_buckets_hidden`: the
    bucketed form with the stand-ins at the two buckets that overlap the backward pass (the last one with T = 0):
    contention + the graph's cross-branch edges."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import dp_corunner_probe as probe
    out = {'k_workgroups': k, 'threads': 512, 'T_us': T, 'ms_per_step_single_graph': single_ms}
    for name, buckets, t_last in (('one_bucket', 1, None), ('three_buckets_hidden', 3, 0.0)):
        us = probe.measure(0, k, T, t_last, 200, n=n, buckets=buckets)
        out[name] = {'ms_per_step': us * 1e-3, 'ratio': us * 1e-3 / single_ms}
    out['one_bucket']['ms_per_step_minus_T'] = out['one_bucket']['ms_per_step'] - T * 1e-3
    out['one_bucket']['ratio_minus_T'] = out['one_bucket']['ms_per_step_minus_T'] / single_ms
    return out


def time_configs(dev, n, reps=200):
    """SURVEY 8(d)'s other timing configurations, full training steps at batch n (four steps per hipGraph replay, like the
    headline; the per-sample k_cpt vector of the adaptive net fed with every step), `reps` steps each after 12 warm-up steps:
    img/s and ms/step.  The headline stays cifar10-ac k_cpt=0."""
    import arch_and_hypers as A
    kv = torch.from_numpy(np.random.default_rng(0).choice(A.k_cpts, n).astype(np.float32)).to(dev)
    table = [
        ('sr_chain8', A.sr_chain(8), 3, lambda net: {}),
        ('ac_k6.4e-8', A.ac_chain(k_cpt=6.4e-8, seed=1234), 3, lambda net: {net.τ: A.τ_ds(0)}),
        ('cr_k0', A.cr_chain(k_cpt=0.0, seed=1234), 3, lambda net: {net.τ: A.τ_cr(0)}),
        ('ac_dyn_kcpt', A.ac_chain(dyn_k_cpt=True, seed=1234), 3, lambda net: {net.τ: A.τ_ds(0), net.k_cpt: kv}),
        ('mnist_sr', A.sr_chain(8), 1, lambda net: {}),
    ]
    out = {}
    for name, make, c0, extra in table:
        try:
            net = make((32, 32, c0), (10,))
            net.to(dev)
            eng = net.engine()
            g = torch.Generator().manual_seed(7)
            eng.ensure_capacity(n, train=True)
            eng.x0[:n].copy_(torch.rand((n, 32, 32, c0), generator=g).to(dev))
            eng.y[:n].copy_(torch.nn.functional.one_hot(torch.randint(0, 10, (n,), generator=g), 10).float().to(dev))
            feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: A.λ_lrn(0)}
            feed.update(extra(net))
            for _ in range(3):
                net.train.run_steps([feed] * 4)         # (four steps per hipGraph replay, as the headline)
            torch.cuda.synchronize()
            ms = time_replays(lambda: net.train.run_steps([feed] * 4), max(1, reps // 4), chunk=5) / 4
            out[name] = {'images_per_s': n / (ms * 1e-3), 'ms_per_step': ms, 'steps_per_graph': 4}
            del net, eng
        except Exception as e:                      # a config that cannot run must not take the headline down
            out[name] = {'error': repr(e)}
    return out


def time_experiment(dev, n, single_ms, reps=100):
    """The reference trains the 8 nets of an experiment -- ac_chain(k_cpt=k), k in k_cpts, scripts/train-nets:81-88 -- one
    after another, each at batch 128 (:159-164).  Here the same 8 nets, each with its own batch, BatchNorm statistics and
    parameters, advance TOGETHER: one hipGraph whose launch j is launch j of every net (lib/_co.py; `train-nets --co-train 8`).
    Aggregate images/s of the 8 co-trained nets beside the serial rate (the single-net headline): the dependency depth
    of a step stays 33 launches, the work per launch grows 8-fold."""
    import arch_and_hypers as A
    from lib._co import CoTrainer, CoGroups
    nets, feeds = [], []
    g = torch.Generator().manual_seed(99)
    for i, k in enumerate(A.k_cpts):
        net = A.ac_chain(k_cpt=k, seed=1234 + i)((32, 32, 3), (10,))
        net.to(dev)
        eng = net.engine()
        eng.ensure_capacity(n, train=True)
        eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g).to(dev))
        eng.y[:n].copy_(torch.nn.functional.one_hot(torch.randint(0, 10, (n,), generator=g), 10).float().to(dev))
        nets.append(net)
        feeds.append({net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: A.λ_lrn(0), net.τ: A.τ_ds(0)})
    K = len(nets)
    co = CoTrainer(nets)
    for _ in range(5):
        co.run(feeds)
    torch.cuda.synchronize()
    ms_one = time_replays(lambda: co.run(feeds), reps)
    # the saturated operating point of the conv bodies: every launch of the joint step (8 nets x 128 images per launch) in
    # situ, eager launches with a HIP-event pair around each -- per family and per launch against the fp32 MFMA peak
    roof = None
    try:
        prog = co._program(n)
        st = torch.cuda.current_stream()
        tot = [0.0] * len(prog['ops'])
        R = 5
        for rp in range(R + 1):
            for e in co.engs:
                e.begin_step(False)
            evs = []
            for op in prog['ops']:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); op(st.cuda_stream); e1.record(st)
                evs.append((e0, e1))
            torch.cuda.synchronize()
            if rp:
                for j, (e0, e1) in enumerate(evs):
                    tot[j] += e0.elapsed_time(e1) / R
        for e in co.engs:
            e.mark_dirty()
        fams = {}
        for op, t in zip(prog['ops'], tot):
            if op.flops:
                f = fams.setdefault(op.what, dict(launches=0, flops=0.0, ms=0.0, members=[]))
                f['launches'] += 1; f['flops'] += op.flops; f['ms'] += t
                tf = op.flops / (t * 1e-3) / 1e12
                f['members'].append({'launch': op.tag, 'us': t * 1e3, 'tflops': tf, 'frac': tf / PEAK_F32_MFMA})
        roof = {}
        for what, f in fams.items():
            tf = f['flops'] / (f['ms'] * 1e-3) / 1e12
            roof[what] = {'bound': 'mfma', 'achieved': tf, 'peak': PEAK_F32_MFMA, 'unit': 'TFLOP/s', 'frac': tf / PEAK_F32_MFMA,
                          'launches_per_joint_step': f['launches'], 'us_per_joint_step': f['ms'] * 1e3,
                          'members': sorted(f['members'], key=lambda m: m['frac'])}
        roof['what'] = ('one group of 8 (one joint step = one launch per layer for all 8 nets, 1 024 images per launch), eager launches, '
                        'HIP events around every launch: the event pairs add the host\'s launch latency to each')
    except Exception as e:
        roof = {'error': repr(e)}
    del co
    # the same 8 nets as 4 groups of 2, each group's joint hipGraph on a stream of its own (lib/_co.py: CoGroups;
    # `train-nets --co-train 8` runs this form): a second hardware queue fills the ramps and drains of the first
    cg = CoGroups.plan(nets, streams=4)
    for _ in range(5):
        cg.run(feeds)
    cg.join()
    torch.cuda.synchronize()
    main_st, per = torch.cuda.current_stream(), []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_st)
        for _ in range(max(1, reps // 5)):
            cg.run(feeds)                           # (the groups free-run against each other; no barrier between steps)
        cg.join()
        e1.record(main_st)
        torch.cuda.synchronize()
        per.append(e0.elapsed_time(e1) / max(1, reps // 5))
    ms = float(np.median(per))
    # ... and TWO experiments of one architecture together (`train-nets cifar10-ac --with cifar10-cr`: the eight ac_chain and
    # the eight cr_chain nets, 16 per joint step): twice the tile-rows per launch for the same 768 resident workgroups --
    # smaller, more equal shares ("where the idle quarter sits")
    g_sizes, g_streams, g_share = [c.K for c in cg.groups], len(cg.streams), cg.share
    two = None
    try:
        nets2, feeds2 = list(nets), list(feeds)
        for i, k in enumerate(A.k_cpts):
            net = A.cr_chain(k_cpt=k, seed=4321 + i)((32, 32, 3), (10,))
            net.to(dev)
            eng = net.engine()
            eng.ensure_capacity(n, train=True)
            eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g).to(dev))
            eng.y[:n].copy_(torch.nn.functional.one_hot(torch.randint(0, 10, (n,), generator=g), 10).float().to(dev))
            nets2.append(net)
            feeds2.append({net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: A.λ_lrn(0), net.τ: A.τ_cr(0)})
        del cg
        cg2 = CoGroups.plan(nets2, streams=4)
        for _ in range(5):
            cg2.run(feeds2)
        cg2.join()
        torch.cuda.synchronize()
        per2 = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_st)
            for _ in range(max(1, reps // 10)):
                cg2.run(feeds2)
            cg2.join()
            e1.record(main_st)
            torch.cuda.synchronize()
            per2.append(e0.elapsed_time(e1) / max(1, reps // 10))
        ms2 = float(np.median(per2))
        two = {'what': 'cifar10-ac + cifar10-cr: 16 nets of one architecture, %s on %d streams' % ([c.K for c in cg2.groups], len(cg2.streams)),
               'nets': len(nets2), 'images_per_s': len(nets2) * n / (ms2 * 1e-3), 'ms_per_joint_step': ms2,
               'speedup_vs_serial': len(nets2) * single_ms / ms2,
               'step_frac_of_mfma_roofline': len(nets2) * n / (ms2 * 1e-3) * F_TRAIN / 1e12 / PEAK_F32_MFMA}
        del cg2
    except Exception as e:
        two = {'error': repr(e)}
    return {'nets': K, 'two_experiments': two, 'what': 'cifar10-ac experiment: ac_chain(k_cpt=k) for the 8 k_cpts, batch %d each, co-trained: %d groups of %d '
                               '(one hipGraph per group: launch j = launch j of its nets) side by side on %d streams'
                               % (n, len(g_sizes), max(g_sizes), g_streams),
            'images_per_s': K * n / (ms * 1e-3), 'ms_per_joint_step': ms, 'ms_per_net_step': ms / K,
            'images_per_s_serial': n / (single_ms * 1e-3), 'speedup_vs_serial': K * single_ms / ms,
            'step_frac_of_mfma_roofline': K * n / (ms * 1e-3) * F_TRAIN / 1e12 / PEAK_F32_MFMA,
            'groups': g_sizes, 'streams': g_streams, 'share': g_share,
            'one_group': {'what': 'all 8 nets in ONE joint hipGraph on one stream', 'images_per_s': K * n / (ms_one * 1e-3),
                          'ms_per_joint_step': ms_one, 'speedup_vs_serial': K * single_ms / ms_one},
            'roofline': roof}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=128, help='per-GPU batch (arch_and_hypers.py:35)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--steps-per-graph', type=int, default=4,
                    help='training steps per hipGraph replay (Engine.run_steps; 1: one graph per step).  A step is a step: '
                         '--steps still counts single training steps')
    ap.add_argument('--streams', action='store_true',
                    help='multi-stream DAG schedule (measured slower under hipGraph: cross-stream edges cost more than the overlap gains)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eval-batch', type=int, default=4096, help='batch of the routed / dense evaluation measurement')
    ap.add_argument('--no-configs', action='store_true', help='skip the SURVEY 8(d) config table (extra.configs)')
    ap.add_argument('--no-experiment', action='store_true', help='skip the co-trained experiment (extra.experiment)')
    ap.add_argument('--no-dp-structure', action='store_true', help='skip the 1-rank RCCL structure measurement (dp_structure)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` by itself: this process becomes the launcher.  It never touches the
        # GPU (no HIP call before the children exist), starts N ranks through torch.distributed.run --
        # exactly the driver's multi-GPU command -- relays their output (rank 0 prints the JSON line)
        # and exits with their code.
        return spawn_ranks(args.gpus)

    # stdout carries ONE line, the JSON record: everything else this process (or a library under it: RCCL prints a version
    # banner through C stdio) writes to file descriptor 1 goes to stderr instead; the record is written to the real
    # stdout at the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    import arch_and_hypers as A
    from lib import _dp
    rank, world = _dp.init()
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d, but the process group has %d rank(s): launch with torch.distributed.run '
                         '--nproc-per-node %d, or run `python bench.py --gpus %d` with WORLD_SIZE unset (it spawns the ranks)'
                         % (args.gpus, world, args.gpus, args.gpus))
    local = _dp.local_device()
    dev = 'cuda:%d' % local
    torch.cuda.set_device(local)

    net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
    net.to(dev)
    eng = net.engine()
    eng.use_graph = not args.no_graph
    eng.multi_stream = args.streams
    _dp.attach(net)
    n = args.batch
    if rank == 0:                              # buffers for the evaluation measurement too: allocated before anything is timed
        eng.ensure_capacity(max(n, args.eval_batch), train=False)
        eng.ensure_capacity(n, train=True)
    x0, y = synthetic(n, rank, dev)
    eng.x0[:n].copy_(x0)
    eng.y[:n].copy_(y)
    feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr',
            net.λ_lrn: A.λ_lrn(0), net.τ: A.τ_ds(0)}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # K training steps per hipGraph replay -- single process AND data parallel (where the step's all-reduce is captured: K
    # steps hold K all-reduces; Engine.run_steps falls back to one step per replay where collectives do not capture): the
    # ~8.6 us the GPU idles between two replays of a one-step graph are paid once per K steps.  `steps` counts training steps.
    k_ok = world == 1 or (eng.allreduce is not None and eng.dp_one_graph and eng.allreduce_capturable and not eng.per_bucket_update)
    spg = max(1, min(args.steps_per_graph, eng.STEPS_MAX)) if (eng.use_graph and not args.streams and k_ok) else 1

    def run_steps(k):
        done = 0
        while spg > 1 and k - done >= spg:
            net.train.run_steps([feed] * spg)
            done += spg
        for _ in range(k - done):
            net.train.run(feed)

    for _ in range(3):                         # eager, capture, first replay of the one-step graph
        net.train.run(feed)
    if spg > 1:
        for _ in range(3):                     # (warm-up, capture, first replay of the K-step graph)
            net.train.run_steps([feed] * spg)
    # The GPU must be BUSY right up to the timed region: its power management drops the clocks within milliseconds of idle and
    # takes milliseconds of work to bring them back -- the first 20-step (10 ms) region after 5 ms of idle reads + 1.5 %, after
    # 20-100 ms + 3 %, after 1 s + 5 %, after 5 s + 10 % (tools/clock_ramp_probe.py, profiles/r06_clock_ramp.txt); a 200-step
    # region dilutes that tenfold.  So everything the host has to do -- Python's garbage collection, which round 6 first put
    # BETWEEN the warm-up and the region: tens of milliseconds of idle -- happens before the warm-up steps, and nothing but
    # the barrier separates them from the timed steps.  (The host enqueues a whole 200-step region in ~6 ms, 0.03 ms per
    # step: it is not the host that the region waits for.)
    # Capturing the graphs above left the GPU idle for ~100 ms, and the driver's W = 5 warm-up steps are 2.4 ms of work: not
    # enough to bring the clocks back (20-step regions then read 0.496-0.502 ms per step against a steady state of 0.486).
    # CLOCK_WARMUP further untimed steps (~50 ms of the same work, reported as config.clock_warmup_steps) run in front of
    # the W warm-up steps; the timed region is still exactly `steps` full training steps.
    import gc
    gc.collect()
    gc.disable()
    run_steps(CLOCK_WARMUP)
    run_steps(max(args.warmup, 3))
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = n * world / (dt / args.steps)
    # data parallel: the collective by itself -- each gradient bucket's all-reduce, timed with HIP events
    # on the launch stream (what the overlap has to hide; 0.3-1.3 MB per bucket over xGMI)
    ar = None
    if world > 1:
        ar = {}
        st_c = torch.cuda.current_stream()
        for name, (lo, hi) in eng.dp_buckets.items():
            view = eng.G[lo:hi]
            for _ in range(3):
                _dp.allreduce_sum_any(view)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st_c)
            for _ in range(20):
                _dp.allreduce_sum_any(view)
            e1.record(st_c)
            e1.synchronize()
            ar[name] = {'bytes': (hi - lo) * 4, 'us': e0.elapsed_time(e1) / 20 * 1e3}

    # Steady state outside the headline region, on EVERY rank (training steps are collective under data
    # parallelism): 400 further replays, HIP events around chunks of ten -> median / mean / p95 step time.
    st_ = torch.cuda.current_stream()
    CH, NCH_ = 10, 40                                         # (an event per replay would break the back-to-back queue)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(NCH_ + 1)]
    CH = CH if spg == 1 else (CH + spg - 1) // spg * spg      # (whole K-step replays per chunk)
    run_steps(3 * spg)
    evs[0].record(st_)
    for k in range(NCH_):
        run_steps(CH)
        evs[k + 1].record(st_)
    barrier()
    per = np.array([evs[k].elapsed_time(evs[k + 1]) / CH for k in range(NCH_)])
    steady = {'steps': CH * NCH_, 'ms_median': float(np.median(per)), 'ms_mean': float(per.mean()),
              'ms_p95': float(np.percentile(per, 95)), 'images_per_s_median': n * world / (float(np.median(per)) * 1e-3),
              'steps_per_graph': spg,
              'what': 'replays after the headline region (rank 0\'s clock), HIP events around chunks of %d steps' % CH}

    # A multi-rank run validates itself: after all the timed steps the replicas must still hold the SAME parameters and
    # momentum accumulators (every rank applied the same all-reduced gradient), whatever form the step took; the
    # BatchNorm moving averages are per-replica state by design (each rank normalises its own 128 images) and are
    # reported, not required to agree.
    dp_check = None
    if world > 1:
        form = (eng.step_graph_form() if eng.use_graph else None) or 'eager'
        if eng.k_step_graph_captured(dp=True):
            form = 'whole, %d steps per graph' % spg
        dP, dA, dS = _dp.max_divergence(eng.P), _dp.max_divergence(eng.A), _dp.max_divergence(eng.S)
        dp_check = {'replicas_identical': dP == 0.0 and dA == 0.0, 'max_abs_param_divergence': dP,
                    'max_abs_momentum_divergence': dA, 'bn_moving_average_divergence': dS,
                    'dp_form': form, 'captured_collective_selftest': getattr(eng, 'dp_selftest', None),
                    'backend': dist.get_backend(), 'steps_before_check': args.warmup + args.steps + 3 + CH * NCH_}
        single = os.environ.get('MPNN_SINGLE_GPU_VALUE')
        if single:
            dp_check['efficiency_vs_single'] = value / (world * float(single))

    dp_structure, cfg_table = None, None
    if world == 1 and not args.no_dp_structure and not args.streams:
        try:
            dp_structure = measure_dp_structure(net, eng, feed, steady['ms_median'])
        except Exception as e:
            dp_structure = {'error': repr(e)}
        try:
            dp_structure['corunner'] = measure_corunner(steady['ms_median'], n)
        except Exception as e:
            dp_structure['corunner'] = {'error': repr(e)}
    if rank == 0 and world == 1 and not args.no_configs:
        cfg_table = time_configs(dev, n)
    experiment = None
    if rank == 0 and world == 1 and not args.no_experiment and not args.streams and not args.no_graph:
        try:
            experiment = time_experiment(dev, n, steady['ms_median'])
        except Exception as e:
            experiment = {'error': repr(e)}

    out = None
    if rank == 0:
        # routed FLOPs/s = images/s x 2 x moc (scripts/train-nets:120), moc from an 'ev' pass
        net.eval({net.x0: eng.x0[:n], net.y: eng.y[:n]})
        moc = float(net.state()[(net, 'moc')].mean())
        # Evaluation mode, forward only (moving-average BatchNorm, hard routing p_ev; scripts/lib/desc.py:10-22):
        # dense at the training batch size, and -- SURVEY 8d "Eval-mode" -- dense vs ROUTED (on-device
        # per-branch compaction, lib/_plan.py:_program_ev) on a statistics-pass-sized batch with the
        # router biases set so that 1/8 of the batch leaves at each of the 8 exits.
        def time_eval(feed_, routed, reps):
            # (median over HIP-event-timed chunks: one hiccup of the host must not colour a 20-call mean)
            for _ in range(4):
                net.eval(feed_, routed=routed)
            torch.cuda.synchronize()
            return time_replays(lambda: net.eval(feed_, routed=routed), reps, chunk=5)
        ev_feed = {net.x0: eng.x0[:n], net.y: eng.y[:n]}
        ev_ms = time_eval(ev_feed, False, 100)
        leaves = [nd.layer for nd in eng.leaves]
        nb_ = args.eval_batch
        xe, ye = synthetic(nb_, 12345, dev)
        x_tr, y_tr = eng.x0[:n].clone(), eng.y[:n].clone()
        eng.x0[:nb_].copy_(xe); eng.y[:nb_].copy_(ye)
        big = {net.x0: eng.x0[:nb_], net.y: eng.y[:nb_]}
        bias_keep = [(ℓ.router.comps[-1].params.b, ℓ.router.comps[-1].params.b.numpy().copy()) for ℓ in net.switches]
        set_exit_fractions(net, big, nb_, [1.0 / 8] * 7)
        d_ms = time_eval(big, False, 20)
        hist_d = [float(l.p_ev.mean()) for l in leaves]
        moc_d = float(net.state()[(net, 'moc')].mean())
        r_ms = time_eval(big, True, 20)
        exit_hist = [float(l.p_ev.mean()) for l in leaves]
        moc_r = float(net.state()[(net, 'moc')].mean())
        blocks_run = sum(h * (k + 1) for k, h in enumerate(exit_hist))       # chain: exit k runs blocks 0..k
        ev = {'images_per_s_forward_dense': n / (ev_ms * 1e-3), 'ms_per_batch': ev_ms, 'batch': n,
              'routed': {'batch': nb_, 'compaction': True,
                         'images_per_s': nb_ / (r_ms * 1e-3), 'ms_per_batch': r_ms,
                         'images_per_s_dense_same_batch': nb_ / (d_ms * 1e-3), 'ms_per_batch_dense': d_ms,
                         'speedup_vs_dense': d_ms / r_ms,
                         'exit_histogram': exit_hist, 'exit_histogram_dense': hist_d,
                         'skipped_block_fraction': 1.0 - blocks_run / max(1, len(eng.blocks)),
                         'moc': moc_r, 'moc_dense': moc_d,
                         'routed_flops_per_s': nb_ / (r_ms * 1e-3) * 2 * moc_r,
                         'router_state': 'synthetic: exit biases calibrated to 1/8 of the batch per exit'},
              'compaction': True}
        eng.x0[:n].copy_(x_tr); eng.y[:n].copy_(y_tr)             # the training batch back in place
        for b_, v_ in bias_keep:                                    # ... and the router biases (replicas stay identical)
            b_.assign(v_)
        # dominant kernel FAMILY (one kernel symbol, or the instantiations of one template): in-situ
        # per-launch HIP-event timing on the launch stream, whole steps run eagerly
        ops = eng.time_step_ops('tr', n, reps=20)
        fam = {}
        for what, tag, fl, t in ops:
            f = fam.setdefault(what, [0, 0.0, 0.0]); f[0] += 1; f[1] += fl; f[2] += t
        conv = [o for o in ops if o[2] > 0]
        dom_name = max((k for k in fam if fam[k][1] > 0), key=lambda k: fam[k][2])
        # the family's mean launch duration: one event pair around its run of consecutive launches
        blocks_t = eng.time_family_blocks('tr', n, reps=20)
        cnt, fl, t_ms = blocks_t[dom_name]
        ach = fl / (t_ms * 1e-3) / 1e12
        # the OTHER conv family beside it, so that the lowest one is always on the line
        other = 'fwd_group' if dom_name != 'fwd_group' else 'bwd_scale'
        roof_other = None
        if other in blocks_t and blocks_t[other][1] > 0:
            c2, f2, t2 = blocks_t[other]
            a2 = f2 / (t2 * 1e-3) / 1e12
            roof_other = {'bound': 'mfma', 'achieved': a2, 'peak': PEAK_F32_MFMA, 'unit': 'TFLOP/s', 'frac': a2 / PEAK_F32_MFMA,
                          'kernel': 'fwd_first_k + fwd_group_k + fwd_ks_k (the forward conv launches)' if other == 'fwd_group'
                                    else 'bwd_scale_k + bwd_level_k (the backward conv launches)',
                          'launches_per_step': c2, 'kernel_ms': t2 / c2, 'kernel_flops': f2 / c2}
        symbol = {'fwd_group': 'fwd_group_k', 'bwd_scale': 'bwd_scale_k<GK,OT,NCH,HASV> + bwd_level_k<GKMASK,OTMASK> (all instantiations: the backward conv launches)',
                  'msconv_fwd': 'conv_k<...,EPI_FWD>'}.get(dom_name, dom_name)
        total_ms = sum(o[3] for o in ops)
        conv_fl, conv_ms = sum(o[2] for o in conv), sum(o[3] for o in conv)
        # HBM-side traffic of the same kernel family per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # passes of THIS command, summarised by tools/summarize_pmc.py into profiles/ (bench.py cannot run
        # the profiler around itself); null when no summary is committed.
        traffic, traffic_src, traffic_stale = None, None, None
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_summary.json')))
            if pm.get('family') == dom_name:
                traffic_stale = pm.get('csrc_sha16') != csrc_sha16()
                traffic_src = '%s; collected %s at source hash %s' % (pm.get('source'), pm.get('collected'), pm.get('csrc_sha16'))
                if not traffic_stale:               # a summary of OLDER kernels is not this run's traffic
                    traffic = pm['traffic_bytes_per_launch']
        except Exception:
            pass
        # launch floor: the same number of launches as a training step, captured and replayed as one hipGraph --
        # (a) of a truly EMPTY kernel (one wave that returns: what a grid boundary costs, comparable across rounds and with
        # the guide's figure), (b) of the smallest kernel of the library that does work (a 1-item slab reduction: two
        # dependent memory round trips; this is what rounds 1-3 reported as `launch_floor`)
        n_launch = len(ops) + (0 if eng.program('tr', n).get('fused_opt') else 1)     # (+ the optimizer when it is a launch of its own)
        st = torch.cuda.current_stream()
        tab = torch.tensor([0, 0, 4, 1, 4, 0], dtype=torch.int32, device=dev)
        buf = torch.zeros(64, device=dev)

        def floor_of(fn):
            fn(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                s_ = torch.cuda.current_stream().cuda_stream
                for _ in range(n_launch):
                    fn(s_)
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(50):
                g.replay()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / 50 * 1e6
        floor_us = floor_of(lambda s_: eng.lib.mpnn_debug_noop(s_))
        small_us = floor_of(lambda s_: eng.lib.mpnn_slab_reduce(buf.data_ptr(), buf[32:].data_ptr(), tab.data_ptr(), 1, s_))

        out = {
            'metric': 'images/sec training CIFAR-10 actor-net', 'value': value, 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'cifar10-ac: ac_chain(k_cpt=0) 8-block actor-routed chain, 32x32x3, 10 classes',
                       'global_batch': n * world, 'per_gpu_batch': n, 'parallelism': 'dp%d' % world, 'rccl_ranks': (dist.get_world_size() if world > 1 else 1),
                       'allreduce': ar,
                       **(dp_check or {}),
                       'hip_graph': bool(eng.use_graph), 'steps_per_graph': spg, 'streams': eng.n_streams if eng.multi_stream else 1,
                       'clock_warmup_steps': CLOCK_WARMUP},
            'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F32_MFMA, 'unit': 'TFLOP/s',
                         'frac': ach / PEAK_F32_MFMA, 'traffic': traffic, 'traffic_source': traffic_src,
                         'traffic_stale': traffic_stale,
                         'kernel': symbol, 'launches_per_step': cnt,
                         'kernel_ms': t_ms / cnt, 'kernel_flops': fl / cnt,
                         # every launch of the family (one per step each), furthest from the roofline first: its own in-situ
                         # HIP-event pair (whole steps run eagerly; the pair adds the host's launch latency, so the members'
                         # times sum to more than launches_per_step x kernel_ms)
                         'members': sorted([{'launch': tag_, 'launches_per_step': 1, 'us': t_ * 1e3,
                                             'tflops': fl_ / (t_ * 1e-3) / 1e12, 'frac': fl_ / (t_ * 1e-3) / 1e12 / PEAK_F32_MFMA}
                                            for what_, tag_, fl_, t_ in ops if what_ == dom_name and fl_ > 0 and t_ > 0],
                                           key=lambda m: m['frac'])},
            'step_frac_of_mfma_roofline': value / world * F_TRAIN / 1e12 / PEAK_F32_MFMA,
            'conv_kernels': {'tflops': conv_fl / (conv_ms * 1e-3) / 1e12, 'sum_ms': conv_ms,
                             'all_launches_sum_ms': total_ms, 'n_launches': len(ops)},
            'routed_flops_per_s': value * 2 * moc, 'moc': moc, 'eval': ev, 'steady_state': steady,
            'dp_structure': dp_structure, 'extra': {'configs': cfg_table, 'experiment': experiment},
            'roofline_forward' if other == 'fwd_group' else 'roofline_backward': roof_other,
            'launch_floor': {'kernels_per_step': n_launch, 'us_per_step': floor_us, 'us_per_kernel': floor_us / n_launch,
                             'what': 'hipGraph of that many EMPTY kernels (one wave, returns at once)'},
            'small_kernel_floor': {'kernels_per_step': n_launch, 'us_per_step': small_us, 'us_per_kernel': small_us / n_launch,
                                   'what': 'hipGraph of that many 1-workgroup kernels with two dependent memory round trips '
                                           '(a 1-item mpnn_slab_reduce): rounds 1-3 reported this as launch_floor'},
        }
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(n)
    if world > 1:
        dist.barrier()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    sys.exit(main() or 0)
