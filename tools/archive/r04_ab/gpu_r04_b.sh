#!/bin/bash
# round 4, call B: stagger sweep of the fused segment kernel, r03 library beside it on the same box, remaining tests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/b; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_tb_resident.py tests/test_gpu_tb_chain.py tests/test_gpu_offload.py tests/test_schsim.py -m gpu -q --maxfail=8 2>&1 | tail -40 > $O/pytest.log; tail -8 $O/pytest.log
echo "== r03 library"; NRLDPC_HIP_LIB=tools/ab/libldpc_hip_r03.so timeout 300 python tools/slot_chain.py 50 | tee $O/slot_chain_r03lib.txt
cd /tmp && NRLDPC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libldpc_hip_r03.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_r03" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 30 0.18 retx > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; f=$(find $O/prof_r03 -name '*kernel_stats.csv' | head -1); cp "$f" $O/chain_kernel_stats_r03lib_retx.csv; head -7 $O/chain_kernel_stats_r03lib_retx.csv | cut -c1-150; rm -rf $O/prof_r03
echo "== unfused"; NRLDPC_HIP_TB_FUSED=0 timeout 300 python tools/slot_chain.py 50 | tee $O/slot_chain_unfused.txt
cd /tmp && NRLDPC_HIP_TB_FUSED=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_unf" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 30 0.18 retx > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; f=$(find $O/prof_unf -name '*kernel_stats.csv' | head -1); cp "$f" $O/chain_kernel_stats_unfused_retx.csv; head -7 $O/chain_kernel_stats_unfused_retx.csv | cut -c1-150; rm -rf $O/prof_unf
for S in 0 4 8 12 16 24; do
  echo "== fused, stagger $S us: $(NRLDPC_HIP_TB_STAGGER_US=$S timeout 300 python tools/slot_chain.py 50 | cut -c1-120)  retx: $(NRLDPC_HIP_TB_STAGGER_US=$S timeout 300 python tools/slot_chain.py 30 0.18 retx | cut -c40-100)"
done | tee $O/stagger_sweep.txt
NRLDPC_HIP_TB_STAGGER_US=8 timeout 300 python tools/tb_latency.py 2>&1 | grep TB | tee $O/tb_latency.txt
