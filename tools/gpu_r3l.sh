#!/bin/bash
# round 3, GPU call l: LDS fence of the BPTT chain's CUs against its consumers
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tools/bench_variants.sh r3l "SBR_TAIL_FENCE_KB=0" "SBR_TAIL_FENCE_KB=124" "SBR_TAIL_FENCE_KB=0 SBR_Y=2" "SBR_TAIL_FENCE_KB=124 SBR_Y=2" "SBR_TAIL_FENCE_KB=100" "SBR_TAIL_FENCE_KB=84" "SBR_TAIL_FENCE_KB=0 SBR_Y=3" "SBR_TAIL_FENCE_KB=124 SBR_Y=3" 2>&1 | tee gpurun_out/r3l_variants.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3l_stats -o s -- python $OLDPWD/bench.py --steps 8 --warmup 3 --repeats 1 --quick > $OLDPWD/gpurun_out/r3l_stats.log 2>&1 )
python tools/trace_gaps.py $(ls gpurun_out/r3l_stats/*kernel_trace.csv | head -1) 3 > gpurun_out/r3l_timeline.txt 2>&1; cat gpurun_out/r3l_timeline.txt | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py tests/test_gpu_dp_two_ranks.py -m gpu -x -q -k "overlapped or c2 or ranks" > gpurun_out/r3l_tests.txt 2>&1; tail -4 gpurun_out/r3l_tests.txt
