#!/bin/bash
# round 3, GPU call S: is the collective's cost a power-state effect?  the probe at the default performance level and at `high`
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
F="amdgpu.ids\|socket.cpp\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
( echo "--- default performance level"; rocm-smi --showperflevel 2>&1 | grep -i "level" | head -2
  PROBE_SHORT=1 timeout 120 python tools/dp_phase_probe.py 2>&1 | grep -v "$F"
  echo "--- rocm-smi --setperflevel high"; rocm-smi --setperflevel high 2>&1 | grep -iv "^=\|^$" | head -3
  rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -4
  PROBE_SHORT=1 timeout 120 python tools/dp_phase_probe.py 2>&1 | grep -v "$F"
  rocm-smi --setperflevel auto > /dev/null 2>&1 ) | tee gpurun_out/r3S_perflevel.txt
