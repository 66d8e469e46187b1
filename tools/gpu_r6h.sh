#!/bin/bash
# round 6, call h: the whole GPU suite on the pruned tree
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q -x --durations=12 > $out/r6h_gputests.txt 2>&1; tail -25 $out/r6h_gputests.txt | cut -c1-250
