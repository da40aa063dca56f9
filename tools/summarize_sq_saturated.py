#!/usr/bin/env python3
"""Joins the three rocprofv3 runs of tools/collect_sq_saturated.sh into one table per conv kernel instantiation and grid
(sums over all its launches in the run):

  us        mean duration of a launch (the kernel-trace run: no counters)
  waves     waves per SIMD (SQ_WAVES / 1 024)
  mfma      share of the SIMD time the matrix pipe is occupied: SQ_INSTS_MFMA x 32 cycles / SIMD-cycles,
            SIMD-cycles = SQ_BUSY_CYCLES / 32 shader engines x 1 024 SIMDs (that quotient / duration = 2.0-2.2 GHz)
  valu      share taken by the OTHER vector instructions at 4.7 cycles each -- fp32 MFMAs and vector instructions share the
            SIMD's ALUs, their times add (tools/probes/mfma_loop_probe.hip)
  idle      1 - mfma - valu: ramp, tail, waits nobody fills
  TF        MFMA instructions x 2 048 FLOP / duration (executed, padding included)
  wait / stall / activ   SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY of SQ_WAVE_CYCLES (per wave), %
  v s l m   instructions per MFMA: other vector, scalar, LDS, vector memory
  cf        SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, %

    python tools/summarize_sq_saturated.py <dir with trace/ pmc1/ pmc2/>
"""
import collections, csv, glob, re, sys

root = sys.argv[1]


def short(name):
    return re.sub(r'^void ', '', name).split('(')[0]


def keep(name):
    return any(k in name for k in ('fwd_group', 'bwd_scale', 'bwd_level', 'fwd_ks', 'fwd_first', 'conv_k', 'wgrad_k'))


dur = collections.defaultdict(list)
for f in glob.glob(root + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r['Kernel_Name'])
        if keep(n):
            dur[(n, r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ('pmc1', 'pmc2'):
    for f in glob.glob(root + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = short(r['Kernel_Name'])
            if keep(n):
                cnt[(n, r['Grid_Size'])][r['Counter_Name']] += float(r['Counter_Value'])

print('%-40s %7s %4s %7s %5s | %5s %5s %5s %6s | %4s %5s %5s | %5s %5s %5s %5s | %4s' % (
    'kernel', 'grid', 'n', 'us', 'waves', 'mfma', 'valu', 'idle', 'TF', 'wait', 'stall', 'activ', 'v', 's', 'l', 'm', 'cf'))
tot = collections.Counter()
for key in sorted(dur, key=lambda k: -sum(dur[k])):
    d, c = dur[key], cnt.get(key, {})
    n = len(d)
    m = max(1.0, c.get('SQ_INSTS_MFMA', 0.0))
    wc = max(1.0, c.get('SQ_WAVE_CYCLES', 0.0))
    simd = max(1.0, c.get('SQ_BUSY_CYCLES', 0.0) / 32 * 1024)
    f_m, f_v = m * 32 / simd, (c.get('SQ_INSTS_VALU', 0.0) - m) * 4.7 / simd
    tot['simd'] += simd; tot['m'] += m * 32; tot['v'] += (c.get('SQ_INSTS_VALU', 0.0) - m) * 4.7; tot['us'] += sum(d)
    pct = lambda a, b: 100.0 * c.get(a, 0.0) / max(1.0, c.get(b, 0.0))
    print('%-40s %7s %4d %7.1f %5.2f | %5.2f %5.2f %5.2f %6.1f | %4.0f %5.0f %5.0f | %5.2f %5.2f %5.2f %5.2f | %4.0f' % (
        key[0][:40], key[1], n, sum(d) / n, c.get('SQ_WAVES', 0.0) / n / 1024, f_m, f_v, 1 - f_m - f_v,
        m * 2048 / (sum(d) * 1e-6) / 1e12,
        pct('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'), pct('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'), pct('SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES'),
        (c.get('SQ_INSTS_VALU', 0) - m) / m, c.get('SQ_INSTS_SALU', 0) / m, c.get('SQ_INSTS_LDS', 0) / m,
        (c.get('SQ_INSTS_VMEM_RD', 0) + c.get('SQ_INSTS_VMEM_WR', 0)) / m, pct('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE')))
print('all conv launches of the run: %d, %.1f ms; SIMD time: mfma %.2f, other vector instructions %.2f, idle %.2f' % (
    sum(len(v) for v in dur.values()), tot['us'] * 1e-3, tot['m'] / tot['simd'], tot['v'] / tot['simd'], 1 - (tot['m'] + tot['v']) / tot['simd']))
