#!/bin/bash
# chain kernels: parity tests of the TX chain, slot timings, rocprofv3 kernel statistics (tools/gpu_chain_prof.sh <tag>)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp; T=${1:-x}
python -m pytest tests/test_gpu_tb_chain.py -m gpu -x -q -k "dlsch or encode or tx or small_tbs" 2>&1 | tail -3
python tools/slot_chain.py 50 | tee $O/slot_chain_$T.txt
python tools/tb_latency.py 2>&1 | tail -12 | tee $O/tb_latency_$T.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_chain_$T" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 30 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; f=$(find $O/prof_chain_$T -name '*kernel_stats.csv' | head -1); cp "$f" $O/chain_kernel_stats_$T.csv; head -8 $O/chain_kernel_stats_$T.csv | cut -c1-150
rm -rf $O/prof_chain_$T
