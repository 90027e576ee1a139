#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_tb_chain.py -m gpu -x -q > $O/pytest_gpu_tb.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu_tb.txt
timeout 120 python tools/host_path_sweep.py 2>&1 | grep chunk | tee $O/host_path_default.txt
timeout 300 python tools/slot_chain.py 50 | tee $O/slot_chain.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_chain" -- python "$GRAFT_REPO_ROOT/tools/slot_chain.py" 30 > "$GRAFT_REPO_ROOT/$O/prof_chain.log" 2>&1
cd "$GRAFT_REPO_ROOT"; f=$(find $O/prof_chain -name '*kernel_stats.csv' | xargs ls -t | head -1); head -6 "$f" | cut -c1-140; cp "$f" $O/chain_kernel_stats.csv
