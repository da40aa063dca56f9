# Round-6 fuzz hunts on the GPU beyond the draws the suite keeps: fresh seed ranges for every GPU fuzzer (tails of the pytest
# output under gpurun_out/r06z/hunt_*.txt).  The CPU counterpart: MPNN_FUZZ_SEED0=5000 MPNN_FUZZ_DRAWS=1200 pytest tests/test_fuzz_ref_graph.py
mkdir -p gpurun_out/r06z
export MPNN_FUZZ_GPU_SEEDS="$(seq -s' ' 200 280)"
timeout 1500 python -m pytest tests/test_fuzz_product_gpu.py -q 2>&1 | tail -n 5 > gpurun_out/r06z/hunt_product.txt
export MPNN_STATE_FUZZ_SEEDS="300 340"
timeout 1500 python -m pytest tests/test_engine_state_fuzz.py -q 2>&1 | tail -n 5 > gpurun_out/r06z/hunt_state.txt
export MPNN_FUZZ_TREE_SEEDS="$(seq -s' ' 100 140)"
timeout 1500 python -m pytest tests/test_fuzz_trees_gpu.py -q 2>&1 | tail -n 5 > gpurun_out/r06z/hunt_trees.txt
export MPNN_FUZZ_ARCH_SEEDS="$(seq -s' ' 100 130)"
timeout 1500 python -m pytest tests/test_fuzz_arch_gpu.py -q 2>&1 | tail -n 5 > gpurun_out/r06z/hunt_arch.txt
export MPNN_FUZZ_WIDTH_SEEDS="$(seq -s' ' 100 140)"
timeout 1500 python -m pytest tests/test_fuzz_exit_widths_gpu.py -q 2>&1 | tail -n 5 > gpurun_out/r06z/hunt_widths.txt
tail -n 2 gpurun_out/r06z/hunt_*.txt
