#!/bin/bash
# round 5: new parity tests (reference-compiled fixtures, caller-supplied CRC predicate), then the whole -m gpu suite
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests/test_gpu_decoder.py tests/test_gpu_encoder.py tests/test_gpu_tb_chain.py -m gpu -q -k "reference or predicate or crc" ) > gpurun_out/r05/pytest_new.txt 2>&1
tail -15 gpurun_out/r05/pytest_new.txt
( time python -m pytest tests -m gpu -q ) > gpurun_out/r05/pytest_gpu.txt 2>&1
tail -8 gpurun_out/r05/pytest_gpu.txt
