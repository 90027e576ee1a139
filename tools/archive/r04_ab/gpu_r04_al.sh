#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/al; mkdir -p $O; export TMPDIR=/tmp
{
echo "== double check-node tasks: extension rows of degree 3..N as tasks of 128 items, two per thread walked together (NRLDPC_HIP_CN_DOUBLE=N; 0 = none)"
timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x 2>&1 | tail -2
timeout 1500 python tools/ab_fast.py d0:NRLDPC_HIP_CN_DOUBLE=0 d5: d4:NRLDPC_HIP_CN_DOUBLE=4 d3:NRLDPC_HIP_CN_DOUBLE=3 d0b:NRLDPC_HIP_CN_DOUBLE=0 d5b:
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab31_double_check_node_tasks.txt
