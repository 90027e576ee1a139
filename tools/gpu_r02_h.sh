#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q 2>&1 | tail -5
{
timeout 120 python tools/host_path_sweep.py 2>&1 | grep -v amdgpu.ids
for ch in 256 512 1024; do
  NRLDPC_HIP_HOST_CHUNK=$ch timeout 120 python tools/host_path_sweep.py 2>&1 | grep -v amdgpu.ids
done
NRLDPC_HIP_HOST_PULL=0 timeout 120 python tools/host_path_sweep.py 2>&1 | grep -v amdgpu.ids | sed "s/^/copy engine: /"
} | tee $O/host_path_pull.txt
