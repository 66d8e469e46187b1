#!/bin/bash
# round 3, GPU call R: the bench-contract and two-rank tests on the final tree (the rest of the suite: profiles/round3_P_gputests.txt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 260 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_dp_two_ranks.py -m gpu -q 2>&1 | tail -6 | cut -c1-200 | tee gpurun_out/r3R_tests.txt
