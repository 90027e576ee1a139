#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/ag; mkdir -p $O; export TMPDIR=/tmp
{
echo "== task records packed in LDS (where there is room: not BG1 Zc=384 R=1/3) instead of scalar loads from the descriptor: desc = NRLDPC_HIP_TREC_LDS=0, lds = default"
timeout 1500 python tools/ab_fast.py desc:NRLDPC_HIP_TREC_LDS=0 lds: desc2:NRLDPC_HIP_TREC_LDS=0 lds2:
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab27_task_records_in_lds.txt
