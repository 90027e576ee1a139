#!/bin/bash
# round 6, call O: final decoder build of the morning (no scratch, double tasks for groups of >= 3 low-degree rows, priority turns; no persistent kernel): tests, A/B vs r05 and vs the stage-1 commit, bench
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$GRAFT_REPO_ROOT/tools/ab
( time timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/r06/pytest_gpu_o.txt 2>&1; tail -4 gpurun_out/r06/pytest_gpu_o.txt
timeout 900 python tools/ab_fast.py r05:NRLDPC_HIP_LIB=$A/libldpc_hip_r05.so stage1:NRLDPC_HIP_LIB=$A/libldpc_hip_r06a.so new: r05b:NRLDPC_HIP_LIB=$A/libldpc_hip_r05.so stage1b:NRLDPC_HIP_LIB=$A/libldpc_hip_r06a.so newb: > gpurun_out/r06/ab_doubles_vs_stage1_vs_r05.txt 2>&1
cat gpurun_out/r06/ab_doubles_vs_stage1_vs_r05.txt
timeout 600 python bench.py --steps 50 > gpurun_out/r06/bench_o.json 2> gpurun_out/r06/bench_o.err
for i in 1 2; do python tools/slot_chain.py 2>/dev/null | tail -1; done
