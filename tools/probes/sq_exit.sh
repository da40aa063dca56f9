cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sq_exit
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O -o sq -- python3 $R/bench.py --no-graph --steps 20 --warmup 5 --no-cpu-baseline --eval-batch 512 > /dev/null 2> $O/err.txt
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/sq_exit/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0][:24]
        if any(k in name for k in ('route_k', 'exit_tail', 'lin_', 'talr', 'pack_k', 'backward_finish')):
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(acc.items()):
    o = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, {n: round(v) for n, v in o.items()})
PY
