cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01v5 -o r01v5 -- python3 $R/bench.py --steps 100 --warmup 20 > $R/gpurun_out/bench_r01v5.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r01v5_fetch -o fetch -- python3 $R/bench.py --steps 30 --warmup 10 > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r01v5_write -o write -- python3 $R/bench.py --steps 30 --warmup 10 > $R/gpurun_out/pmc_write.log 2>&1
cd $R
tail -1 gpurun_out/bench_r01v5.log | cut -c1-900
python tools/analyze_trace.py gpurun_out/prof_r01v5 | head -3
mkdir -p gpurun_out/pmc_all && cp gpurun_out/pmc_r01v5_fetch/*counter_collection.csv gpurun_out/pmc_all/fetch_counter_collection.csv; cp gpurun_out/pmc_r01v5_write/*counter_collection.csv gpurun_out/pmc_all/write_counter_collection.csv
python tools/summarize_pmc.py gpurun_out/pmc_all 'bwd_scale_k<2, 4, 2>'
ls gpurun_out/pmc_r01v5_fetch | head
