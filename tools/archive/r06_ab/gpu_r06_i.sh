#!/bin/bash
# round 6, call I: the build without scratch memory anywhere + persistent launches for one-workgroup-per-CU codes: tests, A/B vs r05, bench
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$GRAFT_REPO_ROOT/tools/ab
( time timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/r06/pytest_gpu_i.txt 2>&1; tail -4 gpurun_out/r06/pytest_gpu_i.txt
timeout 900 python tools/ab_fast.py r05:NRLDPC_HIP_LIB=$A/libldpc_hip_r05.so new: nopersist:NRLDPC_HIP_PERSIST=0 r05b:NRLDPC_HIP_LIB=$A/libldpc_hip_r05.so newb: nopersistb:NRLDPC_HIP_PERSIST=0 > gpurun_out/r06/ab_new_vs_r05.txt 2>&1
cat gpurun_out/r06/ab_new_vs_r05.txt
timeout 600 python bench.py --steps 50 > gpurun_out/r06/bench_i.json 2> gpurun_out/r06/bench_i.err
python tools/slot_chain.py > gpurun_out/r06/slot_chain_i.txt 2>&1; tail -3 gpurun_out/r06/slot_chain_i.txt
NRLDPC_HIP_LIB=$A/libldpc_hip_r05.so python tools/slot_chain.py > gpurun_out/r06/slot_chain_r05.txt 2>&1; tail -3 gpurun_out/r06/slot_chain_r05.txt
