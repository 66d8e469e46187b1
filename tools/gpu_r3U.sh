#!/bin/bash
# round 3, GPU call U: sync collectives on the engine's streams (tools/dp_sync_probe2.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 150 python tools/dp_sync_probe2.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r3U_dp_sync_probe2.txt
