cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o sq -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0][:28]
        if 'fwd_group' in name or 'bwd_scale' in name:
            acc[(name, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for key, c in sorted(acc.items()):
    o = {k: sum(v)/len(v) for k, v in c.items()}
    wc = o.get('SQ_WAVE_CYCLES', 1)
    print('%-28s grid %7s n %3d  wave_cyc %9.0f wait_any %4.1f%% wait_inst %4.1f%% active %4.1f%%  valu %8.0f salu %8.0f mfma %7.0f  valu/mfma %5.1f salu/mfma %5.1f' % (
        key[0], key[1], len(c['SQ_WAVE_CYCLES']), wc, 100*o.get('SQ_WAIT_ANY',0)/wc, 100*o.get('SQ_WAIT_INST_ANY',0)/wc, 100*o.get('SQ_ACTIVE_INST_ANY',0)/wc,
        o.get('SQ_INSTS_VALU',0), o.get('SQ_INSTS_SALU',0), o.get('SQ_INSTS_MFMA',0), o.get('SQ_INSTS_VALU',0)/max(1,o.get('SQ_INSTS_MFMA',1)), o.get('SQ_INSTS_SALU',0)/max(1,o.get('SQ_INSTS_MFMA',1))))
PY
tail -3 gpurun_out/pmc_sq.log | cut -c1-300
