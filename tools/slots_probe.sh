mkdir -p gpurun_out/r06d
for s in 0 4 5; do
  echo "== MPNN_SLOTS_PER_CU=$s"
  MPNN_SLOTS_PER_CU=$s timeout 300 python tools/train_sweep.py 1024 2>&1 | grep -v amdgpu.ids | cut -c1-420
  MPNN_SLOTS_PER_CU=$s timeout 300 python tools/cotrain_probe.py 8 2>&1 | grep "K = 8"
  MPNN_SLOTS_PER_CU=$s timeout 300 python tools/eval_sweep.py 2>&1 | grep "batch   4096"
done
MPNN_SLOTS_PER_CU=4 BATCH=1024 timeout 300 python tools/trace_phases.py 2>&1 | grep "^fwd_group\|^bwd_scale" | head -30
