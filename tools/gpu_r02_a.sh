#!/bin/bash
# round 2, GPU call A: link microbenchmark, the resident submission path under threads, the GPU test suite, a bench line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
L=$GRAFT_REPO_ROOT/openairinterface5g_amd/lib/libldpc_hip.so
nproc > $O/host.txt; grep -c processor /proc/cpuinfo >> $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>&1
timeout 120 tools/ubench/h2d_link.bin > $O/h2d_link.txt 2>&1; echo "h2d_link rc=$?"
run() { echo "$1 T=$2 case=${4:-mix}: $(env $1 timeout 120 ./tests/abi_threads.bin $L $2 ${3:-400} $4 2>&1 | tail -1 | cut -c1-230)"; }
{
  run X=1 1 2000 1
  run X=1 1 2000 0
  run NRLDPC_HIP_SERVER=0 1 1000 1
  for T in 1 4 16 32 64; do run X=1 $T 600; done
  for T in 1 16 32; do run NRLDPC_HIP_SERVER=0 $T 300; done
  run NRLDPC_HIP_SRV_SLOTS=32 32 600
  run NRLDPC_HIP_SRV_SLOTS=16 32 600
} 2>&1 | tee $O/abi_threads.txt
timeout 300 python tests/ldpctest_hip.py -l 8448 -s 10 -n 200 > $O/ldpctest_hip_8448.txt 2>&1; tail -3 $O/ldpctest_hip_8448.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.txt
timeout 300 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
