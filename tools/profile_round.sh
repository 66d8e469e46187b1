#!/bin/bash
# Runs ON THE GPU BOX (via gpurun):  tools/profile_round.sh <tag> [bench args]
# 1. rocprofv3 --kernel-trace --stats of a short bench run    -> gpurun_out/<tag>_stats/
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)       -> gpurun_out/<tag>_pmc_{fetch,write}/
# 3. tools/pmc_summary.py folds 1+2 into                      -> gpurun_out/<tag>_pmc.json
# 4. bench.py (full default run, with cpu_baseline), roofline.traffic taken from 3 -> gpurun_out/<tag>_bench.json
# Copy the summaries you want judged into profiles/ afterwards.
tag=${1:-round}; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
mkdir -p $out
cd $repo
cd /tmp && export TMPDIR=/tmp
short="python $repo/bench.py --steps 8 --warmup 3 --repeats 1 --no-cpu-baseline $@"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- $short > $out/${tag}_stats.log 2>&1
# counter passes serialise kernels: the overlapped tail's consumers cannot run beside the chain there -> SBR_TAIL_OVERLAP=2 (same kernels, one stream)
SBR_TAIL_OVERLAP=2 timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_fetch -o f -- $short > $out/${tag}_pmc_fetch.log 2>&1
SBR_TAIL_OVERLAP=2 timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_write -o w -- $short > $out/${tag}_pmc_write.log 2>&1
cd $repo
python tools/pmc_summary.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_stats > $out/${tag}_pmc.json
python bench.py --pmc-json $out/${tag}_pmc.json "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json
