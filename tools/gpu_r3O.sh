#!/bin/bash
# round 3, GPU call O: kernel timeline of the data-parallel step (one rank, RCCL)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
R=$PWD
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3O_tr -o s -- python $R/bench.py --force-dp --steps 8 --warmup 3 --repeats 1 --quick > $R/gpurun_out/r3O_tr.log 2>&1 )
python tools/trace_gaps.py $(find gpurun_out/r3O_tr -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r3O_dp_timeline.txt 2>&1
cut -c1-160 gpurun_out/r3O_dp_timeline.txt
tail -3 gpurun_out/r3O_tr.log | cut -c1-300
