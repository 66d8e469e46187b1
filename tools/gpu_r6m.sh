#!/bin/bash
# round 6, call m: the batch builder beside the step in flight -- tests, then the training loop with the previous library and this one
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_batch_builder.py tests/test_gpu_train_cli.py -m gpu -q 2>&1 | tail -5
S="--no-cpu-baseline --no-pmc --sustained-seconds 0 --repeats 3 --no-other-configs"
i=0
for v in "SBR_LIB=tools/probes/variants/libsbr_prebb.so" "X=1" "SBR_LIB=tools/probes/variants/libsbr_prebb.so" "X=2"; do
  i=$((i+1))
  env $v python bench.py $S --loop-iters 2000 > $out/r6m_c2_$i.json 2>> $out/r6m.err
  python - "$out/r6m_c2_$i.json" "$v" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2][-24:].ljust(26), d['ms_per_step'], 'loop', d.get('train_loop'))
P
  env $v python tools/bench_train_loop.py --iters 2000 --host-iters 2 > $out/r6m_loop_$i.json 2>> $out/r6m.err
  python - "$out/r6m_loop_$i.json" "$v" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2][-24:].ljust(26), {k: d[k] for k in d if 'native' in k})
P
done
