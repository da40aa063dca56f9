cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sq_types; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INSTS_[A-Z0-9_]*\|SQ_VALU[A-Z0-9_]*\|SQ_ACTIVE_INST[A-Z0-9_]*\|SQ_INST_CYCLES[A-Z0-9_]*\|SQ_THREAD_CYCLES[A-Z0-9_]*\|SQ_WAIT_INST[A-Z0-9_]*" | sort -u > $O/counters.txt
cat $O/counters.txt | tr '\n' ' '
echo
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $O/p1 -o p1 -- python3 $R/tools/ablate_saturated.py 1024 > $O/ops.txt 2> $O/p1.log; tail -n 2 $O/p1.log
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/p2 -o p2 -- python3 $R/tools/ablate_saturated.py 1024 > /dev/null 2> $O/p2.log; tail -n 2 $O/p2.log
