#!/bin/bash
# round 3, GPU call j: the other configs on the round-3 tree, the bf16 ranking test, timelines of C3 / C4
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_bf16_projection.py -m gpu -q > gpurun_out/r3j_bf16.txt 2>&1; tail -8 gpurun_out/r3j_bf16.txt | cut -c1-300
for c in c1 c3 c4 c5 l128; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --no-pmc --sustained-seconds 0 --loop-iters 0 --repeats 3 > gpurun_out/r3j_${c}_bench.json 2> gpurun_out/r3j_${c}_bench.err
  python - <<P
import json
try:
    d=json.loads(open("gpurun_out/r3j_${c}_bench.json").read().strip().splitlines()[-1])
    print("$c", d["ms_per_step"], d["value"], {k: v for k, v in d["phases_us"].items() if k != "note"})
except Exception as e:
    print("$c FAILED", e); print(open("gpurun_out/r3j_${c}_bench.err").read()[-500:])
P
done
for c in c4 c3; do
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3j_${c}_stats -o s -- python $OLDPWD/bench.py --config $c --steps 6 --warmup 2 --repeats 1 --quick > $OLDPWD/gpurun_out/r3j_${c}_stats.log 2>&1 )
f=$(ls gpurun_out/r3j_${c}_stats/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_gaps.py $f 3 > gpurun_out/r3j_${c}_timeline.txt 2>&1; echo "== $c"; cat gpurun_out/r3j_${c}_timeline.txt | cut -c1-150
done
