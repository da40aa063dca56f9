#!/bin/bash
# SQ instruction mix of every SINGLE-OP conv launch (one (block, scale) each: conv_k / conv_pair_k / wgrad_k instantiations) of a
# training step at a saturating batch -- which body executes how many vector instructions per MFMA.
#   bash tools/sq_single_ops.sh [batch] [tag]  ->  gpurun_out/<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
N=${1:-1024}
O=$R/gpurun_out/${2:-sq_ops}
mkdir -p $O
B="python3 $R/tools/ablate_saturated.py $N"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc1 -o p1 -- $B > $O/ops.txt 2> $O/pmc1.log
timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc2 -o p2 -- $B > /dev/null 2> $O/pmc2.log
cd $R
python3 tools/summarize_sq_ops.py $O > $O/summary.txt 2>&1
cat $O/summary.txt | head -90
