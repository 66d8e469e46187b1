#!/bin/bash
# round 6, call x: LDS bank conflicts of the C2 step's kernels (the head's logits phase)
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $repo/gpurun_out/r6x_lds -o p -- python $repo/bench.py --steps 6 --warmup 2 --repeats 1 --quick > $repo/gpurun_out/r6x_lds.log 2>&1
python $repo/tools/lds_conflicts.py $repo/gpurun_out/r6x_lds | tee $repo/gpurun_out/r6x_lds_conflicts.txt
