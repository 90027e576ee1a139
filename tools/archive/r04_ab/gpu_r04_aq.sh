#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/aq; mkdir -p $O; export TMPDIR=/tmp
{
echo "== fused segment kernel compiled for six waves per SIMD (80 VGPRs, 33 spilled): three workgroups per CU for the R = 8/9 shape (NRLDPC_HIP_TB_W6=1) vs two (=0)"
for T in 0 1 0 1; do
  echo "W6=$T: fused slot $(NRLDPC_HIP_TB_W6=$T timeout 300 python tools/slot_chain.py 50 | cut -c40-90)   retx: $(NRLDPC_HIP_TB_W6=$T timeout 300 python tools/slot_chain.py 30 0.18 retx | cut -c40-90)"
done
NRLDPC_HIP_TB_W6=0 timeout 300 python tools/tb_latency.py 2>&1 | grep " 16 TB\| 32 TB\| 64 TB" | sed 's/^/W6=0 /'
timeout 300 python tools/tb_latency.py 2>&1 | grep " 16 TB\| 32 TB\| 64 TB" | sed 's/^/W6=1 /'
NRLDPC_HIP_TB_TRACE=$O/trace.bin timeout 300 python tools/slot_chain.py 3 > /dev/null 2>&1; python tools/wg_trace.py $O/trace.bin | grep "prologue steps\|kernel span\|distinct" ; rm -f $O/trace.bin
timeout 900 python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py -m gpu -q 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids | tee $O/ab_fused_three_workgroups_per_cu.txt
