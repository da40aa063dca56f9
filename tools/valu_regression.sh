#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/valu_reg; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/p -o p -- python3 $R/tools/valu_regression.py run > $O/cases.txt 2> $O/err.log
cd $R; python3 tools/valu_regression.py fit $O | tee $O/fit.txt
