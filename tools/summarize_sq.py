"""SQ counter summary per conv kernel and grid from a rocprofv3 --pmc pass
(SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA).

    python tools/summarize_sq.py <dir with *counter_collection.csv>
"""
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0][:28]
        if 'fwd_group' in name or 'bwd_scale' in name or 'bwd_level' in name or 'fwd_ks' in name:
            acc[(name, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
print('%-28s %8s %4s %11s %9s %10s %7s %10s %9s' % ('kernel', 'grid', 'n', 'wave_cycles', 'wait_any', 'wait_inst', 'active', 'valu/mfma', 'salu/mfma'))
for key, c in sorted(acc.items()):
    o = {k: sum(v) / len(v) for k, v in c.items()}
    wc = o.get('SQ_WAVE_CYCLES', 1); m = max(1.0, o.get('SQ_INSTS_MFMA', 1))
    print('%-28s %8s %4d %11.0f %8.1f%% %9.1f%% %6.1f%% %10.1f %9.1f' % (
        key[0], key[1], len(c['SQ_WAVE_CYCLES']), wc, 100 * o.get('SQ_WAIT_ANY', 0) / wc, 100 * o.get('SQ_WAIT_INST_ANY', 0) / wc,
        100 * o.get('SQ_ACTIVE_INST_ANY', 0) / wc, o.get('SQ_INSTS_VALU', 0) / m, o.get('SQ_INSTS_SALU', 0) / m))
