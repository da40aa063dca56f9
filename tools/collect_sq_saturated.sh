#!/bin/bash
# SQ counters of the conv bodies at SATURATION (K = 8 joint training step; dense evaluation at 4 096 images), eager
# launches.  Three rocprofv3 runs: kernel trace (durations), then two --pmc passes of 8 SQ counters each.
# -> gpurun_out/sq_sat/{trace,pmc1,pmc2}/  and  gpurun_out/sq_sat/summary.txt (copy to profiles/r06_sq_saturated.txt)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/${1:-sq_sat}
mkdir -p $O
W="${WHAT:-cotrain eval}"
B="python3 $R/tools/sq_saturated.py $W"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- $B > $O/trace.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc1 -o p1 -- $B > $O/pmc1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc2 -o p2 -- $B > $O/pmc2.log 2>&1
cd $R
python3 tools/summarize_sq_saturated.py $O > $O/summary.txt 2>&1
head -70 $O/summary.txt; tail -n 3 $O/pmc1.log; tail -n 3 $O/pmc2.log
