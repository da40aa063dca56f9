for c in 2 3 0; do
  echo "== MPNN_LEVEL_WIDE_CAP=$c"
  MPNN_LEVEL_WIDE_CAP=$c timeout 300 python tools/train_sweep.py 128 1024 2>&1 | grep -v amdgpu.ids | grep "^ *[0-9]" | cut -c1-330
  MPNN_LEVEL_WIDE_CAP=$c timeout 300 python tools/cotrain_probe.py 8 2>&1 | grep "K = 8"
done
