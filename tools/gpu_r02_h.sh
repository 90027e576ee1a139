#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
{
for ch in 64 128 192 256 384 512; do
  NRLDPC_HIP_HOST_CHUNK=$ch timeout 120 python tools/host_path_sweep.py 2>&1 | grep -v amdgpu.ids | sed "s/^/pull=1 /"
done
} | tee $O/host_path_pull.txt
