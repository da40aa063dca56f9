#!/bin/bash
# Kernel traces of 8 co-trained nets as ONE joint graph and as 4 groups of 2 side by side -> gpurun_out/final/streams_trace.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/final
mkdir -p $O; rm -rf $O/st1 $O/st4
COGROUPS=1:8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/st1 -o kt -- python3 $R/tools/streams_probe.py ac 2 > $O/st1.log 2> $O/st1.err
COGROUPS=4:4 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/st4 -o kt -- python3 $R/tools/streams_probe.py ac 2 > $O/st4.log 2> $O/st4.err
cd $R
( echo "# rocprofv3 --kernel-trace of tools/streams_probe.py (the last 12 % of each run: the COGROUPS part); tools/overlap_trace.py"
  echo; echo "## 8 nets, ONE joint hipGraph on one stream"; grep "groups on" $O/st1.log; python tools/overlap_trace.py $O/st1 0.12
  echo; echo "## 8 nets, 4 groups of 2, each group's hipGraph on its own stream (hardware queue)"; grep "groups on" $O/st4.log; python tools/overlap_trace.py $O/st4 0.12 ) > $O/streams_trace.txt
rm -rf $O/st1 $O/st4
cat $O/streams_trace.txt
