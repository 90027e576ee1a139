#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
export NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_encstop.so
for rep in 1 2; do
for k in 9 1 3 4 0 19 20 16; do
  ( cd /tmp && NRLDPC_HIP_ENC_STOP=$k timeout 120 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r05/tr$k" -- python "$GRAFT_REPO_ROOT/tools/tx_enc_driver.py" 20 enc > /dev/null 2>&1 )
  echo "stop=$k: $(python tools/prof_kernels.py gpurun_out/r05/tr$k | grep enc_packed | cut -c70-)"; rm -rf gpurun_out/r05/tr$k
done; done | tee gpurun_out/r05/enc_phase_cost.txt
