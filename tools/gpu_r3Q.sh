#!/bin/bash
# round 3, GPU call Q: where the data-parallel step's time goes (one rank): phases alone, with a process group, with collectives
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 200 python tools/dp_phase_probe.py; echo "--- process group without device_id"; PROBE_DEVICE_ID=0 timeout 200 python tools/dp_phase_probe.py
  echo "--- engine on a torch stream of its own"; PROBE_OWN_STREAM=1 timeout 200 python tools/dp_phase_probe.py ) 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r3Q_dp_probe.txt
