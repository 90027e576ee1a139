#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/m; mkdir -p $O; export TMPDIR=/tmp
for L in openairinterface5g_amd/lib/libldpc_hip.so tools/ab/libldpc_hip_pre_u2.so; do
  echo "== $L"
  NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L NRLDPC_HIP_TB_TRACE=$O/trace.bin timeout 300 python tools/slot_chain.py 3 > /dev/null 2>&1; python tools/wg_trace.py $O/trace.bin | grep -v "first 24\|CU index\|differ in"; rm -f $O/trace.bin
  echo "slot: $(NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python tools/slot_chain.py 50 | cut -c40-90)  retx: $(NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python tools/slot_chain.py 30 0.18 retx | cut -c40-90)"
  echo "unfused slot: $(NRLDPC_HIP_TB_FUSED=0 NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python tools/slot_chain.py 50 | cut -c40-90)"
  NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python tools/tb_latency.py 2>&1 | grep " 1 TB\| 16 TB"
done 2>&1 | tee $O/ab_preload.txt
timeout 600 python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py tests/test_gpu_offload.py -m gpu -q 2>&1 | tail -3
