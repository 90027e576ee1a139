#!/bin/bash
# round 5: the eight-way cut (in-library devices, RCCL loopback), bench line hygiene
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests/test_gpu_tb_resident.py tests/test_bench.py -m gpu -q -x -k "eight or rccl or plain" ) > gpurun_out/r05/pytest_eight.txt 2>&1
tail -30 gpurun_out/r05/pytest_eight.txt
