#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/at; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== can two items per thread stand in for waves?  The double-task build (decoder_ab31) at 16 and at 8 waves per workgroup (NRLDPC_HIP_FAST_WAVES), double tasks off (d0) and on (d5); head = committed kernel"
timeout 1500 python tools/ab_fast.py head:NRLDPC_HIP_LIB=$A/libldpc_hip_head.so w16d0:NRLDPC_HIP_LIB=$A/libldpc_hip_dbl.so,NRLDPC_HIP_CN_DOUBLE=0 w16d5:NRLDPC_HIP_LIB=$A/libldpc_hip_dbl.so w8d0:NRLDPC_HIP_LIB=$A/libldpc_hip_dbl.so,NRLDPC_HIP_CN_DOUBLE=0,NRLDPC_HIP_FAST_WAVES=8 w8d5:NRLDPC_HIP_LIB=$A/libldpc_hip_dbl.so,NRLDPC_HIP_FAST_WAVES=8 w12d5:NRLDPC_HIP_LIB=$A/libldpc_hip_dbl.so,NRLDPC_HIP_FAST_WAVES=12
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab34_items_for_waves.txt
