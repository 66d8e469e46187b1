#!/bin/bash
# Builds timing-experiment variants of the pipelined recurrent kernels (X6P_DBG bit mask, see sbr_rec_p.hip) as
# gpurun_variants/libsbr_dbg<N>.so; run on the GPU box with  SBR_LIB=.../libsbr_dbgN.so python bench.py ...
cd "$(dirname "$0")/../../sequence-based-recommendations_amd/csrc" || exit 1
mkdir -p ../../tools/probes/variants
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -DX6P_DBG=$n -c sbr_rec_p.hip -o /tmp/sbr_rec_p_dbg$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/variants/libsbr_dbg$n.so sbr_api.o sbr_rec.o /tmp/sbr_rec_p_dbg$n.o sbr_rec_q.o sbr_rec_cl.o sbr_batch.o sbr_gemm.o sbr_gemm_x6.o sbr_misc.o &
done
wait
ls -la ../../tools/probes/variants/
