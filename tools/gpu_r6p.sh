#!/bin/bash
# round 6, call p: C2 chains, instruction trims -- A/B of libraries on one box: tools/gpu_r6p.sh <libA|-> <libB|-> ...   ('-' = this tree)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
S="--no-cpu-baseline --no-pmc --sustained-seconds 0 --repeats 5 --no-other-configs --loop-iters 0"
i=0
for rep in 1 2; do
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then e="X=1"; else e="SBR_LIB=tools/probes/variants/$v"; fi
  env $e python bench.py $S > $out/r6p_c2_$i.json 2>> $out/r6p.err
  python - "$out/r6p_c2_$i.json" "$v" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels']
print(sys.argv[2].ljust(22), d['ms_per_step'], 'fwd', k['rec_fwd']['us'], 'bwd', k['rec_bwd']['us'], d['repeats']['ms_per_step'])
P
done
done
