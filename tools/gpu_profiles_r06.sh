#!/bin/bash
# Round 6 evidence in one GPU call: the whole -m gpu suite, round 4's evidence set on HEAD (tools/gpu_profiles_r04.sh with this
# round's tag: bench line, rocprofv3 kernel statistics, PMC passes on THIS build -> profiles/hbm_traffic.json, chain kernels and
# their counters, per-segment ABI, host path, micro-benchmarks), the DL half's statistics, the decoder's per-workgroup timelines
# (operating point + fixed work), throughput over a sweep of noise levels, random-input soaks against the oracle.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06/final; mkdir -p $O; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
bash tools/gpu_profiles_r04.sh r06
bash tools/gpu_r05_tx_prof.sh final_tx > /dev/null 2>&1
cp gpurun_out/r05/final_tx/tx_enc_kernel_stats.txt gpurun_out/r05/final_tx/tx_enc_pmc.txt gpurun_out/r05/final_tx/tx_enc_events.txt gpurun_out/r05/final_tx/tb_tx_phases.txt $O/ 2>/dev/null
rm -rf gpurun_out/r05/final_tx; cat $O/tx_enc_kernel_stats.txt
echo "== decoder timelines"; timeout 300 python tools/dec_trace.py 2>/dev/null > $O/dec_trace_headline.txt; grep "kernel span\|workgroup: total\|list scheduling" $O/dec_trace_headline.txt | cut -c1-220
DEC_TRACE_CODE=1,384,23 timeout 300 python tools/dec_trace.py 2>/dev/null > $O/dec_trace_r23.txt
echo "== noise sweep"; timeout 600 python tools/ab_snr.py head: 2>/dev/null | tee $O/snr_sweep.txt
echo "== soaks"; timeout 200 python tools/soak.py 90 777 2>&1 | tail -3 | tee $O/soak_decoder.txt; timeout 120 python tools/soak_enc.py 45 2>&1 | tail -3 | tee $O/soak_encoder.txt
echo "== cut graphs (first transmissions / zero tails)"
for a in 0 1; do echo "NRLDPC_HIP_TB_TRUNC=$a"; NRLDPC_HIP_TB_TRUNC=$a python tools/slot_chain.py 2>&1 | tail -1; done | tee $O/ab_trunc_slot_final.txt
for a in 0 1; do echo "NRLDPC_HIP_TB_LROW=$a"; NRLDPC_HIP_TB_LROW=$a python tools/slot_chain.py 2>&1 | tail -1; done | tee $O/ab_lrow_slot_final.txt
for a in 0 1; do echo "NRLDPC_HIP_CUT=$a"; NRLDPC_HIP_CUT=$a python tools/seg_call_latency.py 3000 2>&1 | tail -1; done | tee $O/seg_call_latency.txt
echo "== BLER curves against the oracle"; timeout 600 python tools/bler_curve.py 2>/dev/null | tee $O/bler_curve.txt
