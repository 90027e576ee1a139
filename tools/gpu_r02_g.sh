#!/bin/bash
# host-buffer path with kernel pulls (NRLDPC_HIP_HOST_PULL) vs the copy-engine pipeline; parity of the host paths; abort test
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_tb_chain.py -m gpu -x -q -s -k "abort or sharded" 2>&1 | tail -6
{
for pull in 1 0; do
  for ch in default 128 256 512 1024; do
    if [ $ch = default ]; then env NRLDPC_HIP_HOST_PULL=$pull timeout 120 python tools/host_path_sweep.py 2>&1 | sed "s/^/pull=$pull /"
    else env NRLDPC_HIP_HOST_PULL=$pull NRLDPC_HIP_HOST_CHUNK=$ch timeout 120 python tools/host_path_sweep.py 2>&1 | sed "s/^/pull=$pull /"; fi
  done
done
} | tee $O/host_path_pull.txt
