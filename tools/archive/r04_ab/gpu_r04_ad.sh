#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ad; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== late draw: the next ticket is requested between a check-node item's two sweeps (cnld) / between a bit-node item's gather and its store (bnld) and looked at when the task is done; q = committed kernel, ld = both"
timeout 1200 python tools/ab_fast.py q: cnld:NRLDPC_HIP_LIB=$A/libldpc_hip_cnld.so bnld:NRLDPC_HIP_LIB=$A/libldpc_hip_bnld.so ld:NRLDPC_HIP_LIB=$A/libldpc_hip_ld.so q2: ld2:NRLDPC_HIP_LIB=$A/libldpc_hip_ld.so
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab25_late_draw.txt
