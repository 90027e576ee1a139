#!/bin/bash
# round 6, VERDICT r05 item 5 (second form): information-column stores issued straight from the input dwords, before the lambda phase
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06/enc_early; mkdir -p $O; export TMPDIR=/tmp
NRLDPC_HIP_ENC_EARLY=1 timeout 600 python -m pytest tests/test_gpu_encoder.py -q -m gpu 2>&1 | tail -3 | tee $O/pytest_early.txt
NRLDPC_HIP_ENC_EARLY=1 timeout 300 python tools/soak_enc.py 30 2>&1 | tail -2 | tee $O/soak_early.txt
for rep in 1 2 3; do
  for arm in 0 1; do
    export NRLDPC_HIP_ENC_EARLY=$arm
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace_${arm}_$rep" -- python "$GRAFT_REPO_ROOT/tools/tx_enc_driver.py" 40 enc > "$GRAFT_REPO_ROOT/$O/trace_${arm}_$rep.log" 2>&1 )
    echo "== EARLY=$arm rep $rep: $(grep encoder $O/trace_${arm}_$rep.log)"
    python tools/prof_kernels.py $O/trace_${arm}_$rep | grep -E "enc_packed"
    rm -rf $O/trace_${arm}_$rep
  done
done 2>&1 | tee $O/ab_enc_early.txt
