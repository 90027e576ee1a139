#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
( time python -m pytest tests/test_gpu_encoder.py tests/test_gpu_tb_chain.py tests/test_schsim.py -m gpu -q -x ) > gpurun_out/r05/pytest_enc_tx.txt 2>&1
tail -4 gpurun_out/r05/pytest_enc_tx.txt
bash tools/gpu_r05_tx_prof.sh $1 > /dev/null 2>&1
cat gpurun_out/r05/$1/tx_enc_kernel_stats.txt gpurun_out/r05/$1/tb_tx_phases.txt | grep -v "^workgroup"
export NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_encstop.so
for k in 9 1 2 3 4 0; do
  ( cd /tmp && NRLDPC_HIP_ENC_STOP=$k timeout 120 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r05/tr$k" -- python "$GRAFT_REPO_ROOT/tools/tx_enc_driver.py" 20 enc > /dev/null 2>&1 )
  echo "stop=$k: $(python tools/prof_kernels.py gpurun_out/r05/tr$k | grep enc_packed | cut -c70-)"; rm -rf gpurun_out/r05/tr$k
done | tee gpurun_out/r05/$1/enc_phase_cost.txt
