#!/bin/bash
# Builds variants of the pipelined recurrent kernels as tools/probes/variants/libsbr_<name>.so (they travel with gpurun; run with
# SBR_LIB=tools/probes/variants/libsbr_<name>.so):   tools/probes/x6p_build.sh name1:"-DX6P_PACK=0" name2:"-DX6P_BWD_LA=2 -DX6P_BWD_NS=3" ...
cd "$(dirname "$0")/../../sequence-based-recommendations_amd/csrc" || exit 1
make -j8 > /dev/null || exit 1
mkdir -p ../../tools/probes/variants
for spec in "$@"; do
  n=${spec%%:*}; f=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize $f -c sbr_rec_p.hip -o /tmp/sbr_rec_p_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/variants/libsbr_$n.so sbr_api.o sbr_rec.o /tmp/sbr_rec_p_$n.o sbr_rec_q.o sbr_rec_cl.o sbr_batch.o sbr_gemm.o sbr_gemm_x6.o sbr_misc.o sbr_sparse.o sbr_cluster.o sbr_head.o ) &
done
wait
ls -la ../../tools/probes/variants/
