#!/bin/bash
# Collect the round's judged profile artefacts on the GPU box into gpurun_out/final/ (copy into profiles/ afterwards).
#   gpurun -- 'bash tools/collect_profiles.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --eval-batch 512 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o write -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --eval-batch 512 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --eval-batch 512 > /dev/null 2> $O/pmc_sq.err
cd $R
mkdir -p $O/pmc_all
cp $O/pmc_fetch/*counter_collection.csv $O/pmc_all/fetch_counter_collection.csv
cp $O/pmc_write/*counter_collection.csv $O/pmc_all/write_counter_collection.csv
python tools/summarize_pmc.py $O/pmc_all bwd_scale $O/pmc_summary.json > $O/pmc_summary.txt
python tools/summarize_sq.py $O/pmc_sq > $O/sq_counters.txt 2>&1
python tools/analyze_trace.py $O/kt > $O/step_timeline.txt
cp $O/kt/*kernel_stats.csv $O/kernel_stats.csv
grep -a '^{' $O/bench.json | tail -1 > $O/bench_line.json
python tools/check_profile_agreement.py $O/kernel_stats.csv $O/bench_line.json > $O/agreement.txt
cat $O/agreement.txt; cat $O/pmc_summary.txt | tail -3; head -3 $O/step_timeline.txt; cut -c1-400 $O/bench_line.json
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/kt/*kernel_trace.csv $O/pmc_all
