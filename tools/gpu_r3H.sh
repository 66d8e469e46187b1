#!/bin/bash
# round 3, GPU call H: the training loop leg by itself (its error message), then the rest of the GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python tools/bench_train_loop.py --iters 200 --host-iters 2 2>&1 | tail -12 | cut -c1-300
echo "---- scatter: atomic form"
SBR_TAIL_SCATTER_LDS=0 timeout 300 python tools/bench_train_loop.py --iters 200 --host-iters 2 2>&1 | tail -4 | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_contract.py::test_counter_passes_sustained_region_and_training_loop_ride_in_the_default_line 2>&1 | tail -15 | cut -c1-250 | tee gpurun_out/r3H_pytest.txt
