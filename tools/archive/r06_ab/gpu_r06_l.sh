#!/bin/bash
# round 6, call L: shapes planned for 20 wave slots per CU (5 per SIMD, kernels at <= 96 VGPRs) vs 16
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 16 20; do NRLDPC_HIP_CU_WAVES=$w python -c "
import openairinterface5g_amd as pkg
pkg.LDPCinit()
print('CU_WAVES=$w', {c: pkg.ldpc.code_info(*c)[4:6] if not isinstance(pkg.ldpc.code_info(*c), dict) else (pkg.ldpc.code_info(*c).get('n_threads'), pkg.ldpc.code_info(*c).get('lds_bytes')) for c in [(1,384,13),(1,384,23),(1,384,89),(1,192,13),(2,208,15),(2,384,15),(2,384,23),(2,64,15)]})
" 2>/dev/null; done
timeout 900 python tools/ab_fast.py w16:NRLDPC_HIP_CU_WAVES=16 w20:NRLDPC_HIP_CU_WAVES=20 w16b:NRLDPC_HIP_CU_WAVES=16 w20b:NRLDPC_HIP_CU_WAVES=20 > gpurun_out/r06/ab_cu_waves.txt 2>&1
cat gpurun_out/r06/ab_cu_waves.txt
for w in 16 20 16 20; do echo "CU_WAVES=$w: $(NRLDPC_HIP_CU_WAVES=$w python tools/slot_chain.py 2>/dev/null | tail -1)"; done | tee gpurun_out/r06/slot_chain_cu_waves.txt
for w in 16 20; do echo "CU_WAVES=$w retx: $(NRLDPC_HIP_CU_WAVES=$w python tools/slot_chain.py 20 0.18 1 2>/dev/null | tail -1)"; done | tee -a gpurun_out/r06/slot_chain_cu_waves.txt
