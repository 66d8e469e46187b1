#!/bin/bash
# round 6, call y: the combining scatter-add of narrow rows -- tests, C1 A/B, C2's stand-alone scatter figure
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide_scatter_forms.py tests/test_gpu_round5.py tests/test_reference_layers.py tests/test_gpu_config_parity.py -m gpu -q -k "not c5_as_benched" > $out/r6y_tests.txt 2>&1; tail -3 $out/r6y_tests.txt
tools/gpu_call.sh r6y "ab:c1:SBR_LIB=tools/probes/variants/libsbr_precomb.so:X=1:SBR_LIB=tools/probes/variants/libsbr_precomb.so:X=2" "timeline:c1"
grep -i "wgrad_kernel\|scat_reduce" $out/r6y_c1_timeline.txt | head -2 | cut -c1-120
for v in "SBR_LIB=tools/probes/variants/libsbr_precomb.so" "X=1"; do env $v python - <<'P'
import numpy as np, bench, sys
from sbr_amd.engine import RNNEngine
for cfg in ("c2", "c1"):
    cell, layers, n_items, loss, ns = bench.CONFIGS[cfg]
    eng = RNNEngine(cell=cell, layers=layers, n_items=n_items, max_length=200, batch_size=256, loss=loss, n_samples=ns)
    hb = bench.synth_batches(1, 256, 200, n_items, ns, "full", 1235)[0]
    eng.set_batch(hb["X"], None, hb["target"], None, hb["pop"], lengths=hb["lengths"])
    eng.train_step()
    print(cfg, "scatter alone:", eng.debug_scatter(20))
    eng.close()
P
done 2>&1 | grep "scatter alone"
