#!/bin/bash
# round 6, call B: persistent workgroups with drawn blocks + LLR prefetch vs one workgroup per block, same library, same box
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/ab_fast.py p0:NRLDPC_HIP_PERSIST=0 p1:NRLDPC_HIP_PERSIST=1 p2:NRLDPC_HIP_PERSIST=2 p0b:NRLDPC_HIP_PERSIST=0 p1b:NRLDPC_HIP_PERSIST=1 p2b:NRLDPC_HIP_PERSIST=2 > gpurun_out/r06/ab_persist.txt 2>&1
cat gpurun_out/r06/ab_persist.txt
timeout 600 python -m pytest tests/test_gpu_decoder.py -x -q 2>&1 | tail -3
