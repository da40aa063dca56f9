"""Per-launch HBM-side traffic of a kernel family from rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE are collected in separate passes: they do not fit one).  Units and gfx950 correction as
in MI355X_MICROARCH.md (HBM section): the counters are KiB; FETCH_SIZE reads half the bytes of
wide coalesced loads on gfx950 and is doubled.

    python tools/summarize_pmc.py <dir with *counter_collection.csv> <family: bwd_scale|fwd_group> [out.json]
"""
import csv, glob, json, os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d, family = sys.argv[1], sys.argv[2]
syms = {'bwd_scale': ('bwd_scale_k<', 'bwd_level_k<'), 'fwd_group': ('fwd_group_k(',)}[family]     # (the level launches belong to the backward family)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
tot = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if any(sym in name for sym in syms):
            key = (name.split('(')[0][:40], r.get('Grid_Size', ''))
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
            tot[r['Counter_Name']].append(float(r['Counter_Value']))
for key, c in sorted(acc.items()):
    out = {k: sum(v) / len(v) for k, v in c.items()}
    print('%-36s grid %8s  launches %4d  FETCH_SIZE x2 %8.3f MB  WRITE_SIZE %8.3f MB' % (
        key[0], key[1], max(len(v) for v in c.values()),
        2 * out.get('FETCH_SIZE', 0) * 1024 / 1e6, out.get('WRITE_SIZE', 0) * 1024 / 1e6))
fetch = 2 * 1024 * sum(tot['FETCH_SIZE']) / max(1, len(tot['FETCH_SIZE']))
write = 1024 * sum(tot['WRITE_SIZE']) / max(1, len(tot['WRITE_SIZE']))
summary = {'family': family, 'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
           'traffic_bytes_per_launch': fetch + write, 'launches_fetch_pass': len(tot['FETCH_SIZE']),
           'launches_write_pass': len(tot['WRITE_SIZE']),
           'collected': time.strftime('%Y-%m-%d'), 'csrc_sha16': __import__('bench').csrc_sha16(),
           'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py`; KiB x 1024, FETCH_SIZE x 2 (gfx950)'}
print(json.dumps(summary))
if len(sys.argv) > 3:
    json.dump(summary, open(sys.argv[3], 'w'), indent=1)
