#!/bin/bash
# round 6, call s3: the sort for the embedding scatter-add behind the head's record instead of beside the head (cluster-chain configurations) -- tests, C3 / C4 / C5 A/B, stamps of the sampled head
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py tests/test_gpu_sparse_update.py tests/test_reference_layers.py tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -q -k "not c5_as_benched" > $out/r6s_tests.txt 2>&1; tail -3 $out/r6s_tests.txt
python tools/samp_prof.py c3 2>&1 | grep "us$"
P=SBR_LIB=tools/probes/variants/libsbr_presort.so
tools/gpu_call.sh r6s "ab:c3:$P:X=1:$P:X=2" "ab:c4:$P:X=1:$P:X=2" "ab:c5:$P:X=1"
