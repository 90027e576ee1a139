#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
for rep in 1 2 3; do for v in A B; do
  ( cd /tmp && NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_tx$v.so timeout 120 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r05/ab$v" -- python "$GRAFT_REPO_ROOT/tools/tx_enc_driver.py" 40 slot > /dev/null 2>&1 )
  echo "$v: $(python tools/prof_kernels.py gpurun_out/r05/ab$v | grep tb_tx_fused | cut -c70-)"; rm -rf gpurun_out/r05/ab$v
done; done | tee gpurun_out/r05/ab_tx_store_trick.txt
