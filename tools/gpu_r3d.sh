#!/bin/bash
# round 3, GPU call d: tail v2 (final scatter chunk as its own launch, slab reduction fused into the W_hid update), full suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
V=tools/probes/variants
tools/bench_variants.sh r3d "SBR_DUMMY=1" "SBR_TAIL_FINAL=0" "SBR_TAIL_FINAL=32" "SBR_TAIL_FUSE_SLABS=0" "SBR_TAIL_FINAL=0 SBR_TAIL_FUSE_SLABS=0" "SBR_TAIL_SCATTER_WGS=128" "SBR_TAIL_SCATTER_WGS=96" "SBR_TAIL_CHUNKS=6" "SBR_LIB=$V/libsbr_bdefla2.so" "SBR_DUMMY=2" 2>&1 | tee gpurun_out/r3d_variants.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3d_stats -o s -- python $OLDPWD/bench.py --steps 8 --warmup 3 --repeats 1 --quick > $OLDPWD/gpurun_out/r3d_stats.log 2>&1 )
f=$(ls gpurun_out/r3d_stats/*/*kernel_trace.csv gpurun_out/r3d_stats/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_gaps.py $f 3 > gpurun_out/r3d_timeline.txt 2>&1; cat gpurun_out/r3d_timeline.txt
timeout 300 python bench.py > gpurun_out/r3d_bench_full.json 2> gpurun_out/r3d_bench_full.err; tail -c 3000 gpurun_out/r3d_bench_full.json; tail -5 gpurun_out/r3d_bench_full.err
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r3d_tests_all.txt 2>&1
tail -40 gpurun_out/r3d_tests_all.txt
