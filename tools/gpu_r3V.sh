#!/bin/bash
# round 3, GPU call V: data-parallel step with sync collectives on the engine's side stream: two-rank tests, bench contract, one-rank rate
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 110 python -m pytest tests/test_gpu_dp_two_ranks.py tests/test_gpu_bench_contract.py -m gpu -q -x -k "two_ranks or one_rank or self_launch or gpus" 2>&1 | tail -6 | cut -c1-220 | tee gpurun_out/r3V_tests.txt
timeout 60 python bench.py --no-cpu-baseline --no-pmc --sustained-seconds 0 --loop-iters 0 --repeats 3 --force-dp 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('--force-dp', d['ms_per_step'], d['data_parallel']['collectives_per_step'], d['data_parallel']['exposed_us_per_step'])" | tee gpurun_out/r3V_dp.txt
