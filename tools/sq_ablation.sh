cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MPNN_HIP_LIB=$R/multipath-nn_amd/libmpnn_hip_ablate.so
for d in 0 2 4 6 7; do
  O=$R/gpurun_out/sq_abl/d$d; mkdir -p $O
  MPNN_CONV_DBG=$d timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc1 -o p1 -- python3 $R/tools/ablate_saturated.py 1024 > $O/ops.txt 2> $O/pmc1.log
  (cd $R; python3 tools/summarize_sq_ops.py $O | grep "conv_k" > $O/summary.txt)
done
cd $R; for d in 0 2 4 6 7; do echo "== dbg $d"; cat gpurun_out/sq_abl/d$d/summary.txt | cut -c1-140; done
