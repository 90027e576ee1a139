#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2 3 4; do for v in A B; do
  echo "$v: $(NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_tx$v.so python tools/slot_chain.py 60 2>/dev/null | tail -1 | cut -c1-110)"
done; done | tee gpurun_out/r05/ab_rx_epilogue_early_count.txt
for v in A B; do echo "$v retx: $(NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_tx$v.so python tools/slot_chain.py 30 0.18 retx 2>/dev/null | tail -1 | cut -c1-110)"; done | tee -a gpurun_out/r05/ab_rx_epilogue_early_count.txt
for v in A B; do echo "$v: $(NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_tx$v.so python tools/tb_latency.py 2>&1 | grep ' 1 TB' | cut -c1-110)"; done | tee -a gpurun_out/r05/ab_rx_epilogue_early_count.txt
