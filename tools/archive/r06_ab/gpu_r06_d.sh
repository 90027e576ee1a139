#!/bin/bash
# round 6, call D: does scratch memory per lane explain the persistent form's constant +20 us?  (base kernel + a scratch probe)
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$GRAFT_REPO_ROOT/tools/ab
AB_ONLY_R13=1 NRLDPC_HIP_PERSIST=0 timeout 900 python tools/ab_fast.py head: s16:NRLDPC_HIP_LIB=$A/libldpc_hip_scratch16.so s64:NRLDPC_HIP_LIB=$A/libldpc_hip_scratch64.so headb: s16b:NRLDPC_HIP_LIB=$A/libldpc_hip_scratch16.so s64b:NRLDPC_HIP_LIB=$A/libldpc_hip_scratch64.so > gpurun_out/r06/ab_scratch_probe.txt 2>&1
cat gpurun_out/r06/ab_scratch_probe.txt
