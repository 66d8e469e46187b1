#!/bin/bash
# round 3, GPU call g: x6r forward (one wave per SIMD, two tiles), short wave-chunks at the scatter-add's end
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_config_parity.py tests/test_reference_layers.py tests/test_gpu_parity.py tests/test_gpu_edge_shapes.py -m gpu -x -q -k "c2 or c1 or reference or one_layer or pipelined or overlapped or fp16 or smallest or two_layer or ragged or unfused or bf16x6 or edge" > gpurun_out/r3g_tests1.txt 2>&1; tail -12 gpurun_out/r3g_tests1.txt
tools/bench_variants.sh r3g "SBR_X6R=0 SBR_TAIL_SHORT_CHUNKS=0" "SBR_X6R=1 SBR_TAIL_SHORT_CHUNKS=0" "SBR_X6R=0" "SBR_X6R=1" "SBR_X6R=0 SBR_TAIL_SHORT_CHUNKS=4" "SBR_X6R=1 SBR_TAIL_SHORT_CHUNKS=4" "SBR_X6R=1 SBR_TAIL_SCATTER_WGS=128" "SBR_X6R=0 SBR_TAIL_SHORT_CHUNKS=0 SBR_Y=2" "SBR_X6R=1 SBR_Y=2" 2>&1 | tee gpurun_out/r3g_variants.txt
python bench.py --config l128 --quick --repeats 3 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l128 x6r', d['ms_per_step'], d.get('phases_us'))"
SBR_X6R=0 python bench.py --config l128 --quick --repeats 3 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l128 x6p', d['ms_per_step'], d.get('phases_us'))"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3g_stats -o s -- python $OLDPWD/bench.py --steps 8 --warmup 3 --repeats 1 --quick > $OLDPWD/gpurun_out/r3g_stats.log 2>&1 )
f=$(ls gpurun_out/r3g_stats/*/*kernel_trace.csv gpurun_out/r3g_stats/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_gaps.py $f 3 > gpurun_out/r3g_timeline.txt 2>&1; cat gpurun_out/r3g_timeline.txt
