#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/aa; mkdir -p $O; export TMPDIR=/tmp
{
echo "== one queue per pass (check-node tasks, then bit-node tickets), finished-task counter (no more polls once the phase is over), record in one load"
NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_chk.so timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x 2>&1 | tail -3
echo "== A/B: r03 = last round's library, off = NRLDPC_HIP_EARLY_BN=0, on = default"
timeout 900 python tools/ab_fast.py r03:NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_r03.so off:NRLDPC_HIP_EARLY_BN=0 on: off2:NRLDPC_HIP_EARLY_BN=0 on2:
echo "== timeline, early bit-node start"
NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_timing.so timeout 300 python tools/task_timing.py
} 2>&1 | grep -v amdgpu.ids | tee $O/decoder_ab22_early_bn_v3.txt
