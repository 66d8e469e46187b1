#!/bin/bash
# round 3, GPU call G: the whole GPU suite on the rebuilt tail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | cut -c1-250 | tee gpurun_out/r3G_pytest.txt
