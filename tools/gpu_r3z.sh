#!/bin/bash
# round 3, GPU call z: kernel timelines of the step with the output layer's gradient kernels on the GEMM stream / on a third stream
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
R=$PWD
for m in 0 1; do
  ( cd /tmp && export TMPDIR=/tmp && SBR_TAIL_OUT_STREAM=$m timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3z_tr$m -o s -- python $R/bench.py --steps 8 --warmup 3 --repeats 1 --quick > $R/gpurun_out/r3z_tr$m.log 2>&1 )
  python tools/trace_gaps.py $(find gpurun_out/r3z_tr$m -name "*kernel_trace.csv" | head -1) 3 > gpurun_out/r3z_timeline$m.txt 2>&1
  echo "=== SBR_TAIL_OUT_STREAM=$m"; cut -c1-150 gpurun_out/r3z_timeline$m.txt
done
