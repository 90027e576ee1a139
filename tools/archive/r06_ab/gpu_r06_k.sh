#!/bin/bash
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/ab_snr.py p0:NRLDPC_HIP_PERSIST=0 p1:NRLDPC_HIP_PERSIST=1 p0b:NRLDPC_HIP_PERSIST=0 p1b:NRLDPC_HIP_PERSIST=1 > gpurun_out/r06/ab_persist_snr.txt 2>&1
cat gpurun_out/r06/ab_persist_snr.txt
