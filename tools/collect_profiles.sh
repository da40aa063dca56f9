#!/bin/bash
# Collect the round's judged profile artefacts on the GPU box into gpurun_out/final/ (copy into profiles/ afterwards).
#   gpurun -- 'bash tools/collect_profiles.sh'         (kernel trace + agreement; the PMC passes: tools/collect_pmc.sh)
# Every step runs under `timeout`: rocprofv3's counter collection was seen to hang inside hipGraphLaunch.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
timeout 600 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-configs --no-dp-structure --no-experiment > $O/bench_under_rocprof.json 2> $O/kt.err
cd $R
python tools/analyze_trace.py $O/kt > $O/step_timeline.txt
cp $O/kt/*kernel_stats.csv $O/kernel_stats.csv
grep -a '^{' $O/bench.json | tail -1 > $O/bench_line.json
python tools/check_profile_agreement.py $O/kernel_stats.csv $O/bench_line.json > $O/agreement.txt
cat $O/agreement.txt; head -3 $O/step_timeline.txt; cut -c1-400 $O/bench_line.json
rm -rf $O/kt/*kernel_trace.csv
