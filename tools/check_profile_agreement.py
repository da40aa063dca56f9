"""Does bench.py's in-situ HIP-event duration of the dominant kernel family agree with rocprofv3's?

    python tools/check_profile_agreement.py profiles/r01_final_kernel_stats.csv profiles/r01_final_bench.json
"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
b = json.load(open(sys.argv[2]))
sym = b['roofline']['kernel'].split('<')[0].split(' ')[0]
names = (sym, 'bwd_level_k') if sym == 'bwd_scale_k' else (sym,)        # the level launches belong to the backward family
sel = [r for r in rows if r['Name'].replace('void ', '').startswith(names)]
calls = sum(int(r['Calls']) for r in sel)
avg = sum(float(r['TotalDurationNs']) for r in sel) / calls / 1e3
print('%s: rocprofv3 %d calls, average %.2f us;  bench.py (HIP events, in situ) %.2f us;  ratio %.3f'
      % (sym, calls, avg, b['roofline']['kernel_ms'] * 1e3, avg / (b['roofline']['kernel_ms'] * 1e3)))
