#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/w; mkdir -p $O; export TMPDIR=/tmp
{
echo "== early bit-node start: counters checked behind the barrier (diagnostic build), decoder + chain tests"
NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_chk.so timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_tb_chain.py -m gpu -q -x 2>&1 | tail -4
echo "== A/B: off = NRLDPC_HIP_EARLY_BN=0 (both barriers waited for), on = default"
timeout 900 python tools/ab_fast.py off:NRLDPC_HIP_EARLY_BN=0 on: off2:NRLDPC_HIP_EARLY_BN=0 on2:
} 2>&1 | tee $O/decoder_ab22_early_bn.txt
