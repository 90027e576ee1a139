#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/aw; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== the first bit-node ticket's record (a scalar load) requested in front of the barrier that ends the check-node phase (prec) vs behind it (head = committed)"
timeout 1500 python tools/ab_fast.py head: prec:NRLDPC_HIP_LIB=$A/libldpc_hip_prec.so head2: prec2:NRLDPC_HIP_LIB=$A/libldpc_hip_prec.so
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab35_ticket_record_before_barrier.txt
