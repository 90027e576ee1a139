#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ah; mkdir -p $O; export TMPDIR=/tmp
{
echo "== last workgroup round of a launch in the latency shape (NRLDPC_HIP_TB_TAIL): 1664-segment slot = 3 full rounds of 512 + 128"
for T in 0 1 0 1; do
  echo "TAIL=$T: fused slot $(NRLDPC_HIP_TB_TAIL=$T timeout 300 python tools/slot_chain.py 50 | cut -c40-90)   retx: $(NRLDPC_HIP_TB_TAIL=$T timeout 300 python tools/slot_chain.py 30 0.18 retx | cut -c40-90)"
done
NRLDPC_HIP_TB_TAIL=0 timeout 300 python tools/tb_latency.py 2>&1 | grep " 1 TB\| 16 TB\| 64 TB" | sed 's/^/TAIL=0 /'
timeout 300 python tools/tb_latency.py 2>&1 | grep " 1 TB\| 16 TB\| 64 TB" | sed 's/^/TAIL=1 /'
timeout 900 python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py tests/test_gpu_offload.py -m gpu -q 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids | tee $O/ab_tail_round_latency_shape.txt
