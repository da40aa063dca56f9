#!/usr/bin/env python3
"""Joins the three rocprofv3 runs of tools/collect_sq_saturated.sh into one table per conv kernel instantiation and grid:
duration (kernel trace), the SQ wait / issue split, instruction mix per MFMA, LDS traffic and conflicts.

    python tools/summarize_sq_saturated.py <dir with trace/ pmc1/ pmc2/>

TFLOP/s = SQ_INSTS_MFMA x 2 048 FLOP (v_mfma_f32_16x16x4_f32; the 32x32x2 form is 4 096) / duration of the un-profiled-counter
trace run; `pipe` = that over 157.3."""
import collections, csv, glob, os, re, sys

root = sys.argv[1]
FLOP_PER_MFMA = float(os.environ.get('FLOP_PER_MFMA', '2048'))


def short(name):
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0]


def keep(name):
    return any(k in name for k in ('fwd_group', 'bwd_scale', 'bwd_level', 'fwd_ks', 'fwd_first', 'conv_k', 'wgrad_k'))


dur = collections.defaultdict(list)
for f in glob.glob(root + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r['Kernel_Name'])
        if keep(n):
            dur[(n, r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('pmc1', 'pmc2'):
    for f in glob.glob(root + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = short(r['Kernel_Name'])
            if keep(n):
                cnt[(n, r['Grid_Size'], r['Workgroup_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))

hdr = '%-52s %8s %5s %8s %7s %6s | %5s %5s %5s | %5s %5s %5s %5s %5s | %6s %6s' % (
    'kernel', 'grid', 'n', 'us', 'TF', 'pipe', 'wait', 'stall', 'activ', 'valu', 'salu', 'lds', 'vmrd', 'vmwr', 'ldscf', 'ldswt')
print(hdr)
print('(wait / stall / activ: SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY of SQ_WAVE_CYCLES, %; valu .. vmwr: instructions per MFMA;')
print(' ldscf: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, %; ldswt: SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES, %)')
rows = []
for key in sorted(dur, key=lambda k: -sum(dur[k])):
    d = dur[key]
    c = {k: sum(v) / len(v) for k, v in cnt.get(key, {}).items()}
    us = sorted(d)[len(d) // 2]
    m = max(1.0, c.get('SQ_INSTS_MFMA', 0.0))
    wc = max(1.0, c.get('SQ_WAVE_CYCLES', 0.0))
    tf = c.get('SQ_INSTS_MFMA', 0.0) * FLOP_PER_MFMA / (us * 1e-6) / 1e12
    pct = lambda a, b: 100.0 * c.get(a, 0.0) / max(1.0, c.get(b, 0.0))
    print('%-52s %8s %5d %8.1f %7.1f %6.2f | %5.1f %5.1f %5.1f | %5.2f %5.2f %5.2f %5.2f %5.2f | %6.1f %6.1f' % (
        key[0][:52], key[1], len(d), us, tf, tf / 157.3,
        pct('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'), pct('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'), pct('SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES'),
        c.get('SQ_INSTS_VALU', 0) / m, c.get('SQ_INSTS_SALU', 0) / m, c.get('SQ_INSTS_LDS', 0) / m,
        c.get('SQ_INSTS_VMEM_RD', 0) / m, c.get('SQ_INSTS_VMEM_WR', 0) / m,
        pct('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'), pct('SQ_WAIT_INST_LDS', 'SQ_WAVE_CYCLES')))
tot = sum(sum(v) for v in dur.values())
print('conv launches in the trace: %d, %.1f ms in all' % (sum(len(v) for v in dur.values()), tot * 1e-3))
