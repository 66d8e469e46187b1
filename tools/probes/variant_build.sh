#!/bin/bash
# Builds variants of ONE source file of the library as tools/probes/variants/libsbr_<name>.so (they travel with gpurun; run with
# SBR_LIB=tools/probes/variants/libsbr_<name>.so):   tools/probes/variant_build.sh name1:sbr_misc:"-DSCAT_DEBUG_NOFLUSH=1" ...
cd "$(dirname "$0")/../../sequence-based-recommendations_amd/csrc" || exit 1
make -j8 > /dev/null || exit 1
mkdir -p ../../tools/probes/variants
OBJS="sbr_api sbr_rec sbr_rec_p sbr_rec_q sbr_rec_cl sbr_batch sbr_gemm sbr_gemm_x6 sbr_misc sbr_sparse sbr_cluster sbr_head"
for spec in "$@"; do
  n=${spec%%:*}; r=${spec#*:}; src=${r%%:*}; f=${r#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize $f -c $src.hip -o /tmp/${src}_$n.o || exit 1
    list=""; for o in $OBJS; do if [ $o = $src ]; then list="$list /tmp/${src}_$n.o"; else list="$list $o.o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/variants/libsbr_$n.so $list ) &
done
wait
ls -la ../../tools/probes/variants/
