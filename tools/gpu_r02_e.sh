#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
for c in 0 64 128 256; do NRLDPC_HIP_HOST_CHUNK=$c timeout 120 python tools/host_path_sweep.py 2>&1 | grep chunk; done | tee $O/host_path_sweep2.txt
NRLDPC_HIP_BAR_PUSH=0 timeout 120 python tools/host_path_sweep.py 2>&1 | grep chunk
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
