#!/bin/bash
# round 6: first transmissions decoded on the rate mode's graph cut behind the last column that received anything (NRLDPC_HIP_TB_TRUNC)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06/trunc; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py tests/test_schsim.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_chain.txt
timeout 300 python tools/soak_tb.py 40 4 2>&1 | tail -2 | tee $O/soak_tb.txt
for rep in 1 2; do
  for arm in 0 1; do
    echo "== NRLDPC_HIP_TB_TRUNC=$arm rep $rep"
    NRLDPC_HIP_TB_TRUNC=$arm python tools/slot_chain.py 2>&1 | tail -4
  done
done 2>&1 | tee $O/ab_trunc_slot.txt
NRLDPC_HIP_TB_TRUNC=1 python tools/tb_latency.py 2>&1 | tail -8 | tee $O/tb_latency_trunc.txt
