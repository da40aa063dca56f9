# A/B of library variants at saturation: training at batch 1024, K = 8 co-training, dense evaluation at 4096; and the headline step
mkdir -p gpurun_out/r06c
for lib in "" $@; do
  tag=${lib:-base}
  L=$PWD/multipath-nn_amd/libmpnn_hip${lib:+_$lib}.so
  echo "== $tag"
  MPNN_HIP_LIB=$L timeout 300 python tools/train_sweep.py 128 1024 2>&1 | grep -v "^ *$" | cut -c1-200
  MPNN_HIP_LIB=$L timeout 300 python tools/cotrain_probe.py 8 2>&1 | grep "K = 8"
  MPNN_HIP_LIB=$L timeout 300 python tools/eval_sweep.py 2>&1 | grep "batch   4096"
done
