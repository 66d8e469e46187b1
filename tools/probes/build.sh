#!/bin/bash
# Builds the micro-benchmarks (run them on the GPU box: tools/probes/<name>_probe); outputs: profiles/round1_h_probes.txt
cd "$(dirname "$0")" || exit 1
for p in mfma_bf16 mfma_f16x3 mfma_bank mfma_share valu_beside_mfma lds_mask gemm_planes split; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o ${p}_probe ${p}_probe.hip || exit 1
done
