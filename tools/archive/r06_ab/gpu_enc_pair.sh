#!/bin/bash
# round 6, VERDICT r05 item 5: two code blocks per 512-thread encoder workgroup (NRLDPC_HIP_ENC_PAIR) against one per workgroup
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06/enc_pair; mkdir -p $O; export TMPDIR=/tmp
# correctness first: the encoder tests with pairing forced on for every batch of >= 2 blocks (odd batches included)
NRLDPC_HIP_ENC_PAIR=2 timeout 600 python -m pytest tests/test_gpu_encoder.py -q -m gpu 2>&1 | tail -3 | tee $O/pytest_pair.txt
NRLDPC_HIP_ENC_PAIR=2 timeout 300 python tools/soak_enc.py 30 2>&1 | tail -2 | tee $O/soak_pair.txt
for rep in 1 2 3; do
  for arm in 0 512; do
    export NRLDPC_HIP_ENC_PAIR=$arm
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace_${arm}_$rep" -- python "$GRAFT_REPO_ROOT/tools/tx_enc_driver.py" 40 enc > "$GRAFT_REPO_ROOT/$O/trace_${arm}_$rep.log" 2>&1 )
    echo "== PAIR=$arm rep $rep: $(grep encoder $O/trace_${arm}_$rep.log)"
    python tools/prof_kernels.py $O/trace_${arm}_$rep | grep -E "enc_packed"
    rm -rf $O/trace_${arm}_$rep
  done
done 2>&1 | tee $O/ab_enc_pair.txt
