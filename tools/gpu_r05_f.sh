#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05/$1
( time python -m pytest tests/test_gpu_tb_chain.py tests/test_schsim.py -m gpu -q -x ) > gpurun_out/r05/pytest_enc_tx.txt 2>&1
tail -4 gpurun_out/r05/pytest_enc_tx.txt
for m in 1 2; do echo "NRLDPC_HIP_TB_CRC_CHUNK=$m"; NRLDPC_HIP_TB_CRC_CHUNK=$m python tools/tx_enc_driver.py 40 slot; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/$1/crc_chunk_ab.txt
( cd /tmp && NRLDPC_HIP_TB_CRC_CHUNK=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r05/$1/trace1" -- python "$GRAFT_REPO_ROOT/tools/tx_enc_driver.py" 30 slot > /dev/null 2>&1 )
( cd /tmp && NRLDPC_HIP_TB_CRC_CHUNK=2 timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r05/$1/trace2" -- python "$GRAFT_REPO_ROOT/tools/tx_enc_driver.py" 30 slot > /dev/null 2>&1 )
for k in 1 2; do python tools/prof_kernels.py gpurun_out/r05/$1/trace$k | grep -E "tb_tx"; done | tee gpurun_out/r05/$1/crc_chunk_kernels.txt
rm -rf gpurun_out/r05/$1/trace1 gpurun_out/r05/$1/trace2
