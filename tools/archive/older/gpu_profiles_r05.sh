#!/bin/bash
# Round 5 evidence in one GPU call: round 4's whole set on HEAD (tools/gpu_profiles_r04.sh with this round's tag) + the DL half
# (encoder / fused TX kernel: kernel statistics, SQ counters, phase clocks) + the whole -m gpu suite with its duration.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05/final; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
bash tools/gpu_profiles_r04.sh r05
bash tools/gpu_r05_tx_prof.sh final_tx > /dev/null 2>&1
cp gpurun_out/r05/final_tx/tx_enc_kernel_stats.txt gpurun_out/r05/final_tx/tx_enc_pmc.txt gpurun_out/r05/final_tx/tx_enc_events.txt gpurun_out/r05/final_tx/tb_tx_phases.txt $O/ 2>/dev/null
cat $O/tx_enc_kernel_stats.txt
if [ -f tools/ab/libldpc_hip_encstop.so ]; then
  export NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_encstop.so
  for k in 9 1 2 3 4 0; do
    ( cd /tmp && NRLDPC_HIP_ENC_STOP=$k timeout 120 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/tr$k" -- python "$GRAFT_REPO_ROOT/tools/tx_enc_driver.py" 20 enc > /dev/null 2>&1 )
    echo "stop=$k: $(python tools/prof_kernels.py $O/tr$k | grep enc_packed | cut -c70-)"; rm -rf $O/tr$k
  done | tee $O/enc_phase_cost.txt
fi
rm -rf gpurun_out/r05/final_tx
