#!/bin/bash
# round 6, call l: C2 without the head's in-kernel gate: stamps against events, the training loop, head / tail tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
S="--no-cpu-baseline --no-pmc --sustained-seconds 0 --repeats 5 --no-other-configs"
for v in "X=1" "SBR_BENCH_EVENTS=1" "X=2" "SBR_BENCH_EVENTS=1"; do
  env $v python bench.py $S --loop-iters 1000 > $out/r6l_c2_$v.json 2> $out/r6l.err; python - "$out/r6l_c2_$v.json" "$v" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2].ljust(22), d['ms_per_step'], d['value'], d['repeats']['ms_per_step'], 'launch_us', r['launch_us'], r.get('launch_us_hip_events_survey'), 'frac', r['frac'], 'loop', (d.get('train_loop') or {}).get('ms_per_iteration'))
P
done
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_bench_contract.py -m gpu -q 2>&1 | tail -3
tools/gpu_call.sh r6l "bench:c1"
