#!/bin/bash
# round 6, call k: C2 with the BPTT chain timed by its own stamps (no event in front of it: the head's in-kernel gate is live)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
S="--no-cpu-baseline --no-pmc --sustained-seconds 0 --loop-iters 0 --repeats 5 --no-other-configs"
for v in "X=1" "SBR_BENCH_EVENTS=1" "X=2" "SBR_BENCH_EVENTS=1"; do
  env $v python bench.py $S > $out/r6k_c2_$v.json 2> $out/r6k.err; python - "$out/r6k_c2_$v.json" "$v" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2].ljust(22), d['ms_per_step'], d['value'], d['repeats']['ms_per_step'], 'launch_us', r['launch_us'], r.get('launch_us_hip_events_survey'), 'frac', r['frac'])
P
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/r6k_stats -o s -- python $OLDPWD/bench.py --steps 10 --warmup 3 --repeats 1 --quick > /dev/null 2>&1 )
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r6k_stats/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:6]: print({k:r[k] for k in ('Name','Calls','AverageNs')} if 'AverageNs' in r else r)
P
timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_config_parity.py -m gpu -q -k "bench or c2 or c5_as_benched_reference" 2>&1 | tail -3
