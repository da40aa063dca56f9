#!/bin/bash
# PMC passes only (rocprofv3 counter collection crashes inside hipGraph replay on this stack: eager launches).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/final
mkdir -p $O
B="python3 $R/bench.py --no-graph --steps 30 --warmup 10 --no-cpu-baseline --no-configs --no-dp-structure --no-experiment --eval-batch 512"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- $B > /dev/null 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o write -- $B > /dev/null 2> $O/pmc_write.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- $B > /dev/null 2> $O/pmc_sq.err
cd $R
mkdir -p $O/pmc_all
cp $O/pmc_fetch/*counter_collection.csv $O/pmc_all/fetch_counter_collection.csv
cp $O/pmc_write/*counter_collection.csv $O/pmc_all/write_counter_collection.csv
python tools/summarize_pmc.py $O/pmc_all bwd_scale $O/pmc_summary.json > $O/pmc_summary.txt
python tools/summarize_sq.py $O/pmc_sq > $O/sq_counters.txt 2>&1
tail -3 $O/pmc_summary.txt; head -12 $O/sq_counters.txt; tail -3 $O/pmc_fetch.err
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_all
