#!/bin/bash
# round 6, call Q: fused segment kernel with the code's tables copied behind the de-matching stores (new) vs HEAD (r06b)
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$GRAFT_REPO_ROOT/tools/ab
for i in 1 2 3; do
  echo "head: $(NRLDPC_HIP_LIB=$A/libldpc_hip_r06b.so python tools/slot_chain.py 2>/dev/null | tail -1)"
  echo "new : $(python tools/slot_chain.py 2>/dev/null | tail -1)"
done | tee gpurun_out/r06/ab_tables_early.txt
echo "head retx: $(NRLDPC_HIP_LIB=$A/libldpc_hip_r06b.so python tools/slot_chain.py 20 0.18 1 2>/dev/null | tail -1)" | tee -a gpurun_out/r06/ab_tables_early.txt
echo "new  retx: $(python tools/slot_chain.py 20 0.18 1 2>/dev/null | tail -1)" | tee -a gpurun_out/r06/ab_tables_early.txt
NRLDPC_HIP_LIB=$A/libldpc_hip_r06b.so python tools/tb_latency.py 2>/dev/null | tail -8 | tee -a gpurun_out/r06/ab_tables_early.txt
python tools/tb_latency.py 2>/dev/null | tail -8 | tee -a gpurun_out/r06/ab_tables_early.txt
timeout 600 python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py -q -x 2>&1 | tail -3
