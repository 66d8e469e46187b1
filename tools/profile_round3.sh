#!/bin/bash
# Runs ON THE GPU BOX (via gpurun):  tools/profile_round3.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of a short run      -> gpurun_out/<tag>_kernel_stats.csv, <tag>_timeline.txt
#   2. per-kernel HBM counters of every kernel (two --pmc passes, tools/pmc_summary.py) -> gpurun_out/<tag>_pmc.json
#   3. the driver's command, `python bench.py` (its own counter passes, sustained region, training loop, CPU leg)
#                                                           -> gpurun_out/<tag>_bench.json
#   4. in-kernel phase counters of the chains               -> gpurun_out/<tag>_rec_phases.txt
#   5. consumer stamps of the overlapped tail, one bench line per other configuration
#   6. pytest -m gpu                                        -> gpurun_out/<tag>_gputests.txt
tag=${1:-round3}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
short="python $repo/bench.py --steps 8 --warmup 3 --repeats 1 --quick"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- $short > $out/${tag}_stats.log 2>&1
cp $(ls $out/${tag}_stats/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
python $repo/tools/trace_gaps.py $(ls $out/${tag}_stats/*kernel_trace.csv | head -1) 3 > $out/${tag}_timeline.txt 2>&1
SBR_TAIL_OVERLAP=2 timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_fetch -o f -- $short > $out/${tag}_pmc_fetch.log 2>&1
SBR_TAIL_OVERLAP=2 timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_write -o w -- $short > $out/${tag}_pmc_write.log 2>&1
cd $repo
python tools/pmc_summary.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_stats > $out/${tag}_pmc.json
timeout 1500 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 1500 $out/${tag}_bench.err
python -c "
import json; d=json.loads(open('$out/${tag}_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','repeats','sustained','train_loop')}); print(d['roofline']); print(d['hbm_traffic']); print(d['phases_us'])"
( timeout 120 python tools/rec_prof.py c2; timeout 120 python tools/tail_prof.py ) > $out/${tag}_rec_phases.txt 2>&1
timeout 120 python tools/tail_trace.py 8 > $out/${tag}_tail_trace.txt 2>&1      # stamps of the tail's consumers against the chain's end
for c in c1 c3 c4 c5 l128; do      # the other configurations: one line each
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-pmc --sustained-seconds 0 --loop-iters 0 --repeats 3 > $out/${tag}_${c}_bench.json 2> $out/${tag}_${c}_bench.err
  python -c "
import json
d=json.loads(open('$out/${tag}_${c}_bench.json').read().strip().splitlines()[-1]); print('$c', d['ms_per_step'], d['value'])"
done
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $out/${tag}_gputests.txt 2>&1
tail -12 $out/${tag}_gputests.txt
