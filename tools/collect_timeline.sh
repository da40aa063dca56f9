#!/bin/bash
# Kernel trace of the bench's training steps -> per-step timeline (gpurun_out/final/step_timeline.txt) + kernel stats.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
O=$R/gpurun_out/final
mkdir -p $O; rm -rf $O/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/quick_step.py 200 > $O/quick_step_under_rocprof.txt 2> $O/kt.err
cd $R
python tools/analyze_trace.py $O/kt > $O/step_timeline.txt
python tools/replay_gaps.py $O/kt >> $O/step_timeline.txt
# the same with four steps per hipGraph replay (Engine.run_steps): the gap between two replays is paid once per four steps
rm -rf $O/kt4
( cd /tmp && SPG=4 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt4 -o kt -- python3 $R/tools/quick_step.py 200 > $O/quick_step4_under_rocprof.txt 2>> $O/kt.err )
python tools/analyze_trace.py $O/kt4 > $O/step_timeline_4steps.txt
python tools/replay_gaps.py $O/kt4 >> $O/step_timeline_4steps.txt
rm -rf $O/kt4
rm -rf $O/kt/*kernel_trace.csv
head -40 $O/step_timeline.txt
