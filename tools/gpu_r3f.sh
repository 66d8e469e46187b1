#!/bin/bash
# round 3, GPU call f: early sort + the scatter-add's own monitor; RNNCluster head; A/B with repeats
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_reference_cluster.py tests/test_gpu_parity.py -m gpu -x -q -k "cluster or overlapped" > gpurun_out/r3f_tests1.txt 2>&1; tail -15 gpurun_out/r3f_tests1.txt
tools/bench_variants.sh r3f "SBR_DUMMY=1" "SBR_TAIL_EARLY_SORT=0 SBR_TAIL_SCATTER_MONITOR=0" "SBR_DUMMY=2" "SBR_TAIL_EARLY_SORT=0 SBR_TAIL_SCATTER_MONITOR=0 SBR_X=2" "SBR_TAIL_SCATTER_MONITOR=0" "SBR_TAIL_EARLY_SORT=0" "SBR_TAIL_SCATTER_WGS=128" "SBR_TAIL_SCATTER_WGS=96" "SBR_TAIL_GEOM=1" "SBR_DUMMY=3" 2>&1 | tee gpurun_out/r3f_variants.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3f_stats -o s -- python $OLDPWD/bench.py --steps 8 --warmup 3 --repeats 1 --quick > $OLDPWD/gpurun_out/r3f_stats.log 2>&1 )
f=$(ls gpurun_out/r3f_stats/*/*kernel_trace.csv gpurun_out/r3f_stats/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_gaps.py $f 3 > gpurun_out/r3f_timeline.txt 2>&1; cat gpurun_out/r3f_timeline.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r3f_tests_all.txt 2>&1
tail -30 gpurun_out/r3f_tests_all.txt
