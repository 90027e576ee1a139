#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ai; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== check-node front end: a' - r' per lane by four sub-dword-addressed 16-bit subtracts + two xors (sdwa) instead of four v_perm + two subtracts (base = committed kernel)"
timeout 1500 python tools/ab_fast.py base: sdwa:NRLDPC_HIP_LIB=$A/libldpc_hip_sdwa.so base2: sdwa2:NRLDPC_HIP_LIB=$A/libldpc_hip_sdwa.so
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab28_sdwa_subtract.txt
