#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/z; mkdir -p $O; export TMPDIR=/tmp
{
echo "== timeline with stamps at the phase switch: s1 check-node queue empty, s2 bit-node ticket drawn, s3 syndrome published / gate decided, s4 ticket record loaded; w = waiting for the task's columns"
echo "-- barriers only"
NRLDPC_HIP_EARLY_BN=0 NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_timing.so timeout 300 python tools/task_timing.py
echo "-- early bit-node start"
NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_timing.so timeout 300 python tools/task_timing.py
} 2>&1 | grep -v amdgpu.ids | tee $O/timeline_phase_switch.txt
