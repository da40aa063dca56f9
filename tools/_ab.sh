for d in 2 3 4; do
  echo "== MPNN_WSPLIT_DIV_WIDE=$d"
  MPNN_WSPLIT_DIV_WIDE=$d python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  MPNN_WSPLIT_DIV_WIDE=$d python tools/profile_ops.py 2>&1 | grep "bwd_scale  " | head -6
done
for d in 3 4; do
  echo "== MPNN_WSPLIT_DIV=$d"
  MPNN_WSPLIT_DIV=$d python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
