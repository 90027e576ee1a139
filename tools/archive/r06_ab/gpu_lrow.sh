#!/bin/bash
# round 6: the fused segment kernel's int8 decoder input kept in LDS (NRLDPC_HIP_TB_LROW) -- VERDICT r05 item 4, possible now that the
# cut graph leaves room
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06/lrow; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tb_chain.py tests/test_gpu_tb_resident.py tests/test_schsim.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_chain.txt
timeout 300 python tools/soak_tb.py 40 4 2>&1 | tail -2 | tee $O/soak_tb.txt
for rep in 1 2 3; do
  for arm in 0 1; do
    echo "== NRLDPC_HIP_TB_LROW=$arm rep $rep"
    NRLDPC_HIP_TB_LROW=$arm python tools/slot_chain.py 2>&1 | tail -1
  done
done 2>&1 | tee $O/ab_lrow_slot.txt
for a in 0 1; do NRLDPC_HIP_TB_LROW=$a NRLDPC_HIP_TB_TRACE=$O/trace$a.bin timeout 300 python tools/slot_chain.py 3 > /dev/null 2>&1; python tools/wg_trace.py $O/trace$a.bin > $O/wg_trace_lrow$a.txt; rm -f $O/trace$a.bin; echo "== LROW=$a"; grep -v "first 24\|CU index" $O/wg_trace_lrow$a.txt | head -9; done
