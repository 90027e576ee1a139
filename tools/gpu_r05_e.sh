#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
( time python -m pytest tests/test_gpu_encoder.py tests/test_gpu_tb_chain.py tests/test_schsim.py tests/test_gpu_tb_resident.py -m gpu -q -x ) > gpurun_out/r05/pytest_enc_tx.txt 2>&1
tail -6 gpurun_out/r05/pytest_enc_tx.txt
bash tools/gpu_r05_tx_prof.sh $1
for t in 128 192 256 320 384 512; do echo "NRLDPC_HIP_ENC_THREADS=$t"; NRLDPC_HIP_ENC_THREADS=$t python tools/tx_enc_driver.py 40 both; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/$1/enc_threads_sweep.txt
for m in 1 2; do echo "NRLDPC_HIP_TB_CRC_CHUNK=$m"; NRLDPC_HIP_TB_CRC_CHUNK=$m python tools/tx_enc_driver.py 40 slot; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/$1/crc_chunk_ab.txt
