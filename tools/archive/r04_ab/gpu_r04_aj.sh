#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/aj; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== kernel code size vs the 64 KB instruction cache: nop1 = no first-pass specialisation of the check-node bodies (messages initialised to zero instead): 69 KB -> 46 KB of code per kernel"
timeout 1500 python tools/ab_fast.py base: nop1:NRLDPC_HIP_LIB=$A/libldpc_hip_nop1.so base2: nop12:NRLDPC_HIP_LIB=$A/libldpc_hip_nop1.so
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab29_code_size.txt
