#!/bin/bash
# round 3, GPU call X: __graft_entry__.smoke() on the tree as committed
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r3X_smoke.txt
