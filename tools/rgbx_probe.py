#!/usr/bin/env python3
"""Upper bound of "block 0's pyramid scales as dense RGBX maps" (round-4 verdict, item 2b) on the FORWARD launches, before
building the producer: the three forward group launches whose small member reads the 3-channel image through the strided
pick (`fwd_group_k<SMALL>`: scalar loads, shift arithmetic) are pointed at pre-built dense [n, H, W, 4] maps instead
(aligned float4 pixels, shift 0, channel 3 = 0 against a zero weight slot) -- same kernels, same results.  Prints the
step time and the first forward launches' in-situ times both ways."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd')]
import numpy as np, torch, arch_and_hypers as A

n = 128


def build(rgbx):
    net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
    eng = net.engine()
    g = torch.Generator().manual_seed(0)
    eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g)); eng.y[:n].zero_(); eng.y[:n, 0] = 1
    if rgbx:
        eng.x4 = {}
        for sh in (1, 2, 3):
            t = torch.zeros((n, 32 >> sh, 32 >> sh, 4), device=eng.dev)
            t[..., :3] = eng.x0[:n, ::1 << sh, ::1 << sh, :]
            eng.x4[sh] = t
        eng.rgbx_probe = True
        eng._progs.clear()
    feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0}
    return net, eng, feed


def step_us(net, feed, reps=400):
    for _ in range(8): net.train.run(feed)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    K = reps // 10
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    evs[0].record(st)
    for k in range(K):
        for _ in range(10): net.train.run(feed)
        evs[k + 1].record(st)
    torch.cuda.synchronize()
    return float(np.median([evs[k].elapsed_time(evs[k + 1]) / 10 for k in range(K)])) * 1e3

a, b = build(False), build(True)
# same results: the forward pass of both engines
a[0].train.run(a[2]); b[0].train.run(b[2]); torch.cuda.synchronize()
for ba, bb in zip(a[1].blocks, b[1].blocks):
    for sa, sb in zip(ba.s, bb.s):
        assert torch.equal(sa[:n], sb[:n])
print('forward maps identical with the dense RGBX operand')
for rnd in range(2):
    for name, (net, eng, feed) in (('strided x0 (shipped)', a), ('dense RGBX maps', b)):
        print('%-22s step %.1f us' % (name, step_us(net, feed)), flush=True)
for name, (net, eng, feed) in (('strided x0 (shipped)', a), ('dense RGBX maps', b)):
    ops = eng.time_step_ops('tr', n, reps=20)
    eng.mark_dirty()
    print(name)
    for what, tag, fl, ms in ops[:5]:
        print('   %-12s %-44s %7.1f us' % (what, tag, ms * 1e3))
