#!/bin/bash
# encoder / TX kernels: parity first (every code, both kernels, reference-compiled fixtures, chains), then the profile
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
( time python -m pytest tests/test_gpu_encoder.py tests/test_gpu_tb_chain.py tests/test_schsim.py -m gpu -q -x ) > gpurun_out/r05/pytest_enc_tx.txt 2>&1
tail -6 gpurun_out/r05/pytest_enc_tx.txt
bash tools/gpu_r05_tx_prof.sh $1
