#!/bin/bash
# round 3, GPU call e: non-uniform time chunks of the overlapped tail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
V=tools/probes/variants
tools/bench_variants.sh r3e "SBR_DUMMY=1" "SBR_TAIL_GEOM=1" "SBR_TAIL_GEOM=2" "SBR_TAIL_GEOM=3.5" "SBR_TAIL_SCATTER_WGS=96" "SBR_TAIL_SCATTER_WGS=128" "SBR_TAIL_SMALL_K=64" "SBR_TAIL_SMALL_K=64 SBR_TAIL_SMALL_SLABS=32" "SBR_TAIL_FUSE_SLABS=0" "SBR_LIB=$V/libsbr_bdefla2.so" "SBR_DUMMY=2" 2>&1 | tee gpurun_out/r3e_variants.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3e_stats -o s -- python $OLDPWD/bench.py --steps 8 --warmup 3 --repeats 1 --quick > $OLDPWD/gpurun_out/r3e_stats.log 2>&1 )
f=$(ls gpurun_out/r3e_stats/*/*kernel_trace.csv gpurun_out/r3e_stats/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_gaps.py $f 3 > gpurun_out/r3e_timeline.txt 2>&1; cat gpurun_out/r3e_timeline.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3e_tests_all.txt 2>&1
tail -30 gpurun_out/r3e_tests_all.txt
