for cfg in "base:0" "base:1" "wsw:0" "wsw:1"; do
  lib=${cfg%%:*}; wt=${cfg##*:}
  L=$PWD/multipath-nn_amd/libmpnn_hip$( [ $lib = base ] || echo _$lib ).so
  echo "== lib $lib  MPNN_FWD_WIDE_TRAIN=$wt"
  MPNN_HIP_LIB=$L MPNN_FWD_WIDE_TRAIN=$wt timeout 300 python tools/cotrain_probe.py 8 2>&1 | grep "K = 8"
  [ $wt = 0 ] && MPNN_HIP_LIB=$L timeout 300 python tools/eval_sweep.py 2>&1 | grep "batch   4096\|batch   8192"
done
