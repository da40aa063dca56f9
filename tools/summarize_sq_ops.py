#!/usr/bin/env python3
"""Per (kernel instantiation, grid) of tools/sq_single_ops.sh, in dispatch order: launches, mean duration, MFMA-pipe
occupancy from the counters (MFMA x 32 cycles / SIMD-cycles, SIMD-cycles = SQ_BUSY_CYCLES / 32 shader engines x 1 024),
instructions per MFMA, and the per-launch op table of the run beside it."""
import collections, csv, glob, re, sys
root = sys.argv[1]
short = lambda n: re.sub(r'^void ', '', n).split('(')[0]
cnt, order, dur = collections.defaultdict(lambda: collections.defaultdict(list)), [], collections.defaultdict(list)
for sub in ('pmc1', 'pmc2'):
    for f in glob.glob(root + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
        for r in rows:
            n = short(r['Kernel_Name'])
            if not any(k in n for k in ('conv_k', 'conv_pair_k', 'wgrad_k', 'fwd_first', 'fwd_ks')):
                continue
            key = (n, r['Grid_Size'])
            if key not in cnt:
                order.append(key)
            cnt[key][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] in ('SQ_INSTS_MFMA',):
                dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
print('%-52s %8s %4s %8s %6s | %5s %5s %5s | %5s %5s %5s %5s %5s' % ('kernel', 'grid', 'n', 'us', 'pipe', 'wait', 'stall', 'activ', 'valu', 'salu', 'lds', 'vmem', 'smem'))
for key in order:
    c = {k: sum(v) / len(v) for k, v in cnt[key].items()}
    m = max(1.0, c.get('SQ_INSTS_MFMA', 0.0))
    wc = max(1.0, c.get('SQ_WAVE_CYCLES', 0.0))
    simd_cycles = c.get('SQ_BUSY_CYCLES', 0.0) / 32 * 1024
    d = dur[key]
    print('%-52s %8s %4d %8.1f %6.3f | %5.1f %5.1f %5.1f | %5.2f %5.2f %5.2f %5.2f %5.2f' % (
        key[0][:52], key[1], len(d), sum(d) / max(1, len(d)), m * 32 / max(1.0, simd_cycles),
        100 * c.get('SQ_WAIT_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        c.get('SQ_INSTS_VALU', 0) / m, c.get('SQ_INSTS_SALU', 0) / m, c.get('SQ_INSTS_LDS', 0) / m,
        (c.get('SQ_INSTS_VMEM_RD', 0) + c.get('SQ_INSTS_VMEM_WR', 0)) / m, c.get('SQ_INSTS_SMEM', 0) / m))
print()
print(open(root + '/ops.txt').read())
