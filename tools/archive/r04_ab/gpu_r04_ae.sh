#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ae; mkdir -p $O; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/tools/ab
{
echo "== records that carry what a task would look up: q = committed kernel; ld = + bit-node late draw; rc = ld + a bit-node ticket's column records in its record (no f_coltbl read in front of the list); rr = rc + a check-node task's row records in its record (no f_rowtbl read in front of the edge table)"
timeout 1500 python tools/ab_fast.py q:NRLDPC_HIP_LIB=$A/libldpc_hip_q.so ld:NRLDPC_HIP_LIB=$A/libldpc_hip_norc.so rc: rr:NRLDPC_HIP_LIB=$A/libldpc_hip_cnrr.so q2:NRLDPC_HIP_LIB=$A/libldpc_hip_q.so ld2:NRLDPC_HIP_LIB=$A/libldpc_hip_norc.so rc2: rr2:NRLDPC_HIP_LIB=$A/libldpc_hip_cnrr.so
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab26_record_carried_lookups.txt
