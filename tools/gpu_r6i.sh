#!/bin/bash
# round 6, call i: round-6 tests (foreign kernel beside the chains / the head), bench contract, C5's reference initialisation on three
# arithmetics (per-array errors -> profiles), params_twin_vs_oracle of every comparison, C5's chains at 128 rows, the driver's command
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_bench_contract.py -m gpu -q > $out/r6i_tests_1.txt 2>&1; tail -15 $out/r6i_tests_1.txt | cut -c1-300
rm -f $out/r6i_c5_arith.jsonl $out/r6i_parity.jsonl
for v in "X=1" "SBR_X6_F16=0 SBR_X6_F16_BWD=0 SBR_GEMM_F16=0 SBR_WGRAD_F16=0" "SBR_TEST_FLAGS=16"; do
  env $v SBR_PARITY_LOG=$out/r6i_c5_arith.jsonl timeout 900 python -m pytest tests/test_gpu_config_parity.py -m gpu -q -k "c5_as_benched_reference" 2>&1 | tail -2 | cut -c1-200
done
SBR_PARITY_LOG=$out/r6i_parity.jsonl timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py tests/test_golden.py -m gpu -q -k "not c5_as_benched" 2>&1 | tail -2
timeout 300 python tools/cl_prof.py c5 128 > $out/r6i_cluster_phases_c5_b128.txt 2>&1; cat $out/r6i_cluster_phases_c5_b128.txt | cut -c1-420
timeout 900 python bench.py > $out/r6i_bench.json 2> $out/r6i_bench.err; tail -c 600 $out/r6i_bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r6i_bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','vs_baseline')}); print(d.get('roofline')); print({k:(v.get('ms_per_step'), (v.get('cpu_baseline') or {}).get('value')) for k,v in (d.get('other_configs') or {}).items()}); print(d.get('cpu_baseline'))
P
