#!/bin/bash
# round 4, call A: fused UL-SCH segment kernel + host LLR / device HARQ modes: tests, slot timings both ways, kernel stats
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04/a; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q --maxfail=8 -x 2>&1 | tail -40 > $O/pytest.log; tail -15 $O/pytest.log
for F in 1 0; do
  echo "== slot chain, NRLDPC_HIP_TB_FUSED=$F"
  NRLDPC_HIP_TB_FUSED=$F timeout 300 python tools/slot_chain.py 50 | tee $O/slot_chain_fused$F.txt
  NRLDPC_HIP_TB_FUSED=$F timeout 300 python tools/slot_chain.py 20 0.18 retx | tee $O/slot_chain_retx_fused$F.txt
  NRLDPC_HIP_TB_FUSED=$F timeout 300 python tools/tb_latency.py 2>&1 | grep TB | tee $O/tb_latency_fused$F.txt
  cd /tmp && NRLDPC_HIP_TB_FUSED=$F timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_chain$F" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 30 > /dev/null 2>&1
  cd "$GRAFT_REPO_ROOT"; f=$(find $O/prof_chain$F -name '*kernel_stats.csv' | head -1); cp "$f" $O/chain_kernel_stats_fused$F.csv; head -8 $O/chain_kernel_stats_fused$F.csv | cut -c1-150
  rm -rf $O/prof_chain$F
done
echo "== host LLRs"; timeout 600 python tools/slot_chain_host.py 10 2>&1 | tee $O/slot_chain_host.json | tail -80
echo "== bench"; timeout 600 python bench.py --steps 30 --no-strong > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
