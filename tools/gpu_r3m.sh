#!/bin/bash
# round 3, GPU call m: tail v3 -- chains fence their CUs, sort beside the forward chain, output-layer work on a third stream
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py tests/test_gpu_dp_two_ranks.py tests/test_gpu_train_cli.py -m gpu -x -q -k "overlapped or c2 or ranks or learns or end_to_end" > gpurun_out/r3m_tests.txt 2>&1; tail -5 gpurun_out/r3m_tests.txt
tools/bench_variants.sh r3m "SBR_DUMMY=1" "SBR_TAIL_FENCE_KB=0 SBR_TAIL_EARLY_SORT=0 SBR_TAIL_OUT_STREAM=0" "SBR_DUMMY=2" "SBR_TAIL_EARLY_SORT=0" "SBR_TAIL_OUT_STREAM=0" "SBR_TAIL_FENCE_KB=0" "SBR_TAIL_FENCE_KB=0 SBR_TAIL_EARLY_SORT=0 SBR_TAIL_OUT_STREAM=0 SBR_Y=2" "SBR_TAIL_SCATTER_WGS=192" "SBR_TAIL_SCATTER_WGS=64" "SBR_DUMMY=3" 2>&1 | tee gpurun_out/r3m_variants.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3m_stats -o s -- python $OLDPWD/bench.py --steps 8 --warmup 3 --repeats 1 --quick > $OLDPWD/gpurun_out/r3m_stats.log 2>&1 )
python tools/trace_gaps.py $(ls gpurun_out/r3m_stats/*kernel_trace.csv | head -1) 3 > gpurun_out/r3m_timeline.txt 2>&1; cat gpurun_out/r3m_timeline.txt | cut -c1-150
